for T in 262144 524288 1048576 2097152 4194304; do
for rep in 1 2; do
IGD_TAIL_THRESHOLD=$T python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('tail_threshold %8d  %8.1f Mrays/s  literal %8.1f  trav1 %6.1f shade %6.1f trav2 %6.1f tail %5.1f' % ($T, d['value'], d['literal_config']['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
done; done
