cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ad; mkdir -p $O
bash tools/ab_env.sh 20 "IGD_SHADOW_OVERLAP=0" "-" > $O/ab_shadow_headline.log 2>&1; cat $O/ab_shadow_headline.log
for n in 8 2; do for e in "IGD_SHADOW_OVERLAP=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; for rep in 1 2; do
  echo -n "[as-rank-of $n] [$e] "; env $E python bench.py --steps 20 --warmup 5 --as-rank-of $n --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('%8.1f Mrays/s  %.3f ms/step trav1 %.1f shade %.1f trav2 %.1f tail %.1f' % (d['value'], d['ms_per_step'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
done; done; done 2>&1 | tee $O/ab_shadow_rankof.log
for e in "IGD_SHADOW_OVERLAP=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base 2>&1 | sed "s/^/[$e] /"; done | tee $O/ab_shadow_principled.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
for e in "IGD_SHADOW_OVERLAP=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base 2>&1 | sed "s/^/[$e] /"; done | tee $O/ab_shadow_standin.log
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -3
