cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
for e in "-" "IGD_TAIL_THRESHOLD=0" "IGD_TAIL_THRESHOLD=65536" "IGD_TAIL_THRESHOLD=262144" "IGD_TAIL_SPLIT=3" "IGD_TAIL_SPLIT=12"; do
  E=$e; [ "$e" = "-" ] && E=""
  env $E bash tools/ab_scene.sh scenes/many_point_lights.json 32 base 2>&1 | head -1 | sed "s/^/[$e] /"
done > $O/mpl_tail.log; cat $O/mpl_tail.log
IGD_TAIL_DEBUG=1 timeout 300 python bench.py --scene scenes/many_point_lights.json --steps 32 --warmup 32 --no-cpu-baseline --no-literal-config 2>&1 | grep "\[tail\]" | tail -12
