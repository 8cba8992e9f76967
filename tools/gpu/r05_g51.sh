cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ao; mkdir -p $O
for v in base tailpp0 tailpp14; do
  if [ $v = base ]; then unset IGD_LIBRARY; else export IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_$v.so; fi
  for rep in 1 2; do echo -n "[as-rank-of 8] $v "; python bench.py --steps 20 --warmup 5 --as-rank-of 8 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('%8.1f Mrays/s  %.3f ms/step trav1 %.1f shade %.1f trav2 %.1f tail %.2f' % (d['value'], d['ms_per_step'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
done; done 2>&1 | tee $O/ab_tailquorum_rankof8.log
unset IGD_LIBRARY
bash tools/ab.sh 20 base tailpp0 tailpp14 > $O/ab_tailquorum_headline.log 2>&1; cat $O/ab_tailquorum_headline.log
