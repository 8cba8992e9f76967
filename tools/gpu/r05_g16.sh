cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
for sc in scenes/diamond_scene_principled.json:32 /tmp/standin_1m_div/standin.json:16 scenes/many_point_lights.json:32; do
  S=${sc%%:*}; N=${sc##*:}
  for t in 1048576 524288 262144 131072; do
    IGD_TAIL_THRESHOLD=$t bash tools/ab_scene.sh $S $N base 2>&1 | sed "s|^|[$(basename $S) thr $t] |"
  done
done > $O/tail_thr_full.log; cat $O/tail_thr_full.log
