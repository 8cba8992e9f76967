cd $GRAFT_REPO_ROOT
O=gpurun_out/r05an; mkdir -p $O
bash tools/ab_scene.sh scenes/many_point_lights.json 32 nopre base > $O/ab_prefetch3_mpl.log 2>&1; cat $O/ab_prefetch3_mpl.log
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 nopre base > $O/ab_prefetch3_principled.log 2>&1; cat $O/ab_prefetch3_principled.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 nopre base > $O/ab_prefetch3_standin.log 2>&1; cat $O/ab_prefetch3_standin.log
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 > $O/tests.log 2>&1; tail -2 $O/tests.log
