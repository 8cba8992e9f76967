#!/bin/bash
# tools/ab_env.sh <steps> "<ENV=.. ENV=..>" ...   ("-" = none): bench.py under each environment, twice
STEPS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
for E in "$@"; do
  EE=$E; [ "$E" = "-" ] && EE=""
  env $EE timeout 300 python $ROOT/bench.py --steps $STEPS --warmup 16 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['stage_ms_rank0']
print('%-60s %8.1f Mrays/s   trav1 %7.1f  shade %7.1f  trav2 %7.1f  tail %6.1f' % ('$E', d['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
done
done
