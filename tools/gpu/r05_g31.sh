cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ac; mkdir -p $O
bash tools/ab.sh 20 base tocc2 tocc1 > $O/ab_tocc_headline.log 2>&1; cat $O/ab_tocc_headline.log
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base tocc2 tocc1 > $O/ab_tocc_principled.log 2>&1; cat $O/ab_tocc_principled.log
bash tools/ab_scene.sh scenes/many_point_lights.json 32 base tocc2 tocc1 > $O/ab_tocc_mpl.log 2>&1; cat $O/ab_tocc_mpl.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base tocc2 tocc1 > $O/ab_tocc_standin.log 2>&1; cat $O/ab_tocc_standin.log
