#!/bin/bash
# Kernel-variant experiments: builds ignis_amd/lib/var/libig_device_hip_<name>.so with extra -D flags.
# usage: tools/build_variant.sh <name> [-DIG_TRAV_OCC=4 -DIG_LDS_STACK=20 ...]      run with IGD_LIBRARY=<that file>
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/ignis_amd/lib/var/$NAME
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-parameter -I$ROOT/include"
sched() { case $1 in traverse) echo "-mllvm -amdgpu-sched-strategy=max-memory-clause";; shade|photon|tail) echo "-mllvm -amdgpu-sched-strategy=max-ilp";; esac; }
for f in traverse shade photon tail comm device; do
  S=$(sched $f); [ -n "${NO_SCHED:-}" ] && S=""
  /opt/rocm/bin/hipcc $FLAGS $S "$@" -c "$ROOT/ignis_amd/csrc/device/$f.hip" -o "$OUT/$f.o" &
done
for f in traverse tail; do
  S=$(sched $f); [ -n "${NO_SCHED:-}" ] && S=""
  /opt/rocm/bin/hipcc $FLAGS $S "$@" -DIG_QNODE=1 -c "$ROOT/ignis_amd/csrc/device/$f.hip" -o "$OUT/${f}_q8.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OUT"/*.o -o "$ROOT/ignis_amd/lib/var/libig_device_hip_$NAME.so"
rm -rf "$OUT"
echo "$ROOT/ignis_amd/lib/var/libig_device_hip_$NAME.so"
