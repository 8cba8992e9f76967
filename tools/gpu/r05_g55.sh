cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ar; mkdir -p $O
for rep in 1 2; do for e in IGD_SIDE_PRIORITY=low -; do E=$e; [ "$e" = "-" ] && E=""
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('%-24s %8.1f Mrays/s  literal %8.1f  tail %.1f' % ('$e', d['value'], d['literal_config']['value'], s['ms_tail']))"
done; done 2>&1 | tee $O/priority20.log
for e in IGD_SIDE_PRIORITY=low -; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base 2>&1 | sed "s/^/[$e] /"; done | tee $O/priority_principled.log
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 > $O/tests.log 2>&1; tail -2 $O/tests.log
