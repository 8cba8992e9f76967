cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ah; mkdir -p $O
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
for e in "IGD_WORK_SHARDS=1" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base 2>&1 | sed "s/^/[$e] /"; done | tee $O/ab_shards3_standin.log
bash tools/ab_env.sh 20 "IGD_WORK_SHARDS=1" "-" > $O/ab_shards3_headline.log 2>&1; cat $O/ab_shards3_headline.log
for e in "IGD_WORK_SHARDS=1" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base 2>&1 | sed "s/^/[$e] /"; done | tee $O/ab_shards3_principled.log
for e in "IGD_WORK_SHARDS=1" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/many_point_lights.json 32 base 2>&1 | sed "s/^/[$e] /"; done | tee $O/ab_shards3_mpl.log
