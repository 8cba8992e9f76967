cd $GRAFT_REPO_ROOT
O=gpurun_out/r05al; mkdir -p $O
bash tools/ab.sh 20 nopre base > $O/ab_prefetch_headline.log 2>&1; cat $O/ab_prefetch_headline.log
bash tools/ab_scene.sh scenes/many_point_lights.json 32 nopre base > $O/ab_prefetch_mpl.log 2>&1; cat $O/ab_prefetch_mpl.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 > $O/tests.log 2>&1; tail -2 $O/tests.log
