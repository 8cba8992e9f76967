#!/bin/bash
# Kernel-variant experiments: builds ignis_amd/lib/var/libig_device_hip_<name>.so with extra -D flags.
# usage: [ONLY="shade tail"] tools/build_variant.sh <name> [-DIG_TRAV_OCC=4 -DIG_LDS_STACK=20 ...]      run with IGD_LIBRARY=<that file>
# ONLY: the translation units the flags concern (device/<unit>.hip; traverse and tail come in both node formats); the others are taken
# from the in-tree build (ignis_amd/lib/*.o, `make` first).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/ignis_amd/lib
OUT=$LIB/var/$NAME
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-parameter -I$ROOT/include"
sched() { case $1 in traverse) echo "-mllvm -amdgpu-sched-strategy=max-memory-clause";; shade|photon|tail) echo "-mllvm -amdgpu-sched-strategy=max-ilp";; esac; }
ALL="traverse raysort shade photon tail comm device"
for f in $ALL; do
  S=$(sched $f); [ -n "${NO_SCHED:-}" ] && S=""
  if [ -z "${ONLY:-}" ] || [[ " $ONLY " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS $S "$@" -c "$ROOT/ignis_amd/csrc/device/$f.hip" -o "$OUT/$f.o" &
    if [ $f = traverse ] || [ $f = tail ]; then
      /opt/rocm/bin/hipcc $FLAGS $S "$@" -DIG_QNODE=1 -c "$ROOT/ignis_amd/csrc/device/$f.hip" -o "$OUT/${f}_q8.o" &
    fi
  else
    cp "$LIB/$f.o" "$OUT/$f.o"
    if [ $f = traverse ] || [ $f = tail ]; then cp "$LIB/${f}_q8.o" "$OUT/${f}_q8.o"; fi
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OUT"/*.o -o "$LIB/var/libig_device_hip_$NAME.so" -ldl
rm -rf "$OUT"
echo "$LIB/var/libig_device_hip_$NAME.so"
