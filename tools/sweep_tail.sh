#!/bin/bash
# Runs ON the GPU box: bench.py for several IGD_TAIL_THRESHOLD values.  usage: tools/sweep_tail.sh <steps> <warmup> <threshold> ...
STEPS=$1; WARM=$2; shift; shift
for T in "$@"; do
for rep in 1 2; do
IGD_TAIL_THRESHOLD=$T python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('steps $STEPS tail_threshold %8d  %8.1f Mrays/s  trav1 %7.1f shade %7.1f trav2 %7.1f tail %6.1f' % ($T, d['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
done; done
