cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "standin or config5 or principled or blend or plastic or bump or class" -n 4 > $O/pytest_standin.log 2>&1
echo "pytest rc=$?" >> $O/pytest_standin.log
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base > $O/ab_principled.log 2>&1
IGD_SHADE_CLASSES=0 bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base > $O/ab_principled_oneKernel.log 2>&1
bash tools/standin_quick.sh r05a 1000000 16 "-" "IGD_SHADE_CLASSES=0" > $O/standin1M.log 2>&1
tail -3 $O/pytest_standin.log; cat $O/ab_principled.log $O/ab_principled_oneKernel.log $O/standin1M.log
