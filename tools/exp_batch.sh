for rep in 1 2; do
for b in 268435456 536870912; do
  IGD_BATCH_RAYS=$b BENCH_CAPACITY=$b timeout 600 python bench.py --steps 192 --warmup 32 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('batch $b  %8.1f Mrays/s  trav1 %7.1f shade %7.1f trav2 %7.1f tail %6.1f' % (d['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
done; done
