cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -2
bash tools/ab.sh 20 prev base > $O/ab_seed_headline.log 2>&1; cat $O/ab_seed_headline.log
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 prev base > $O/ab_seed_principled.log 2>&1; cat $O/ab_seed_principled.log
bash tools/ab_scene.sh scenes/many_point_lights.json 32 prev base > $O/ab_seed_mpl.log 2>&1; cat $O/ab_seed_mpl.log
