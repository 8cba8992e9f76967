cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -2
bash tools/ab.sh 20 prev base > $O/ab_kinds_headline.log 2>&1; cat $O/ab_kinds_headline.log
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 prev base > $O/ab_kinds_principled.log 2>&1; cat $O/ab_kinds_principled.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 prev base > $O/ab_kinds_standin.log 2>&1; cat $O/ab_kinds_standin.log
bash tools/ab_scene.sh scenes/many_point_lights.json 32 prev base > $O/ab_kinds_mpl.log 2>&1; cat $O/ab_kinds_mpl.log
