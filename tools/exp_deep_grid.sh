python tools/make_standin_scene.py /tmp/standin16 --triangles 16000000 > /dev/null 2>&1
for rep in 1 2; do for g in auto 8 64 768; do
  if [ $g = auto ]; then unset IGD_DEEP_GRID; else export IGD_DEEP_GRID=$g; fi
  python bench.py --scene /tmp/standin16/standin.json --steps 16 --warmup 16 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('deep grid %-5s %8.1f Mrays/s   trav1 %7.1f  shade %7.1f  trav2 %7.1f' % ('$g', d['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary']))"
done; done
unset IGD_DEEP_GRID
for g in auto 768; do if [ $g = auto ]; then unset IGD_DEEP_GRID; else export IGD_DEEP_GRID=$g; fi
python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('diamond deep grid $g', d['value'])"
done
