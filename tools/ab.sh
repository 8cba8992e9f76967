#!/bin/bash
# A/B of kernel variants on the GPU box: tools/ab.sh <steps> <name> [<name> ...]   ("base" = the in-tree library)
# prints value (Mrays/s) and the stage times of `python bench.py --steps <steps>` for each, twice
STEPS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then unset IGD_LIBRARY; else export IGD_LIBRARY=$ROOT/ignis_amd/lib/var/libig_device_hip_$n.so; fi
  timeout 300 python $ROOT/bench.py --steps $STEPS --warmup 16 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['stage_ms_rank0']
print('%-14s %8.1f Mrays/s   trav1 %7.1f  shade %7.1f  trav2 %7.1f  tail %6.1f  launch %.3f ms' % ('$n', d['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail'], d['roofline']['avg_launch_ms']))"
done
done
