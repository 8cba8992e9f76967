cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 2>&1 | tail -2
bash tools/ab_env.sh 20 "IGD_SKIP_MISSES=0" "-" > $O/ab_skipmiss_headline.log 2>&1; cat $O/ab_skipmiss_headline.log
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base > $O/ab_principled.log 2>&1; cat $O/ab_principled.log
