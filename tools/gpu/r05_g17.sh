cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k "standin or principled or blend or plastic or quantised or sphere" 2>&1 | tail -2
bash tools/ab.sh 20 base > $O/ab_headline.log 2>&1; cat $O/ab_headline.log
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base > $O/ab_principled.log 2>&1; cat $O/ab_principled.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base > $O/ab_standin.log 2>&1; cat $O/ab_standin.log
bash tools/ab_scene.sh scenes/many_point_lights.json 32 base > $O/ab_mpl.log 2>&1; cat $O/ab_mpl.log
