cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
for e in "-" "IGD_SHADE_CLASSES=0"; do
  E=$e; [ "$e" = "-" ] && E=""
  env $E bash tools/ab_scene.sh scenes/many_point_lights.json 32 base 2>&1 | sed "s/^/[$e] /"
done > $O/mpl_classes.log; cat $O/mpl_classes.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_mpl -o stats -- python bench.py --scene scenes/many_point_lights.json --steps 32 --warmup 32 --no-cpu-baseline --no-literal-config > /dev/null 2> $O/stats.err
python tools/prof_summary.py stats "$(find $O/stats_mpl -name '*.db' | head -1)" | head -14 ; find $O -name "*.db" -delete
