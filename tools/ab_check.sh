#!/bin/bash
# A/B of kernel variants WITH a parity check: tools/ab_check.sh <steps> <name> [<name> ...]   ("base" = the in-tree library)
# per variant: __graft_entry__.smoke() (primary hits bit-exact, radiance vs the oracle) and the config-size counter test, then tools/ab.sh
STEPS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for n in "$@"; do
  if [ "$n" = base ]; then unset IGD_LIBRARY; else export IGD_LIBRARY=$ROOT/ignis_amd/lib/var/libig_device_hip_$n.so; fi
  echo "== $n: $(python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1 | cut -c1-90)"
  python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "counters or deep or incoherent or config" 2>&1 | tail -1
done
unset IGD_LIBRARY
bash tools/ab.sh "$STEPS" "$@"
