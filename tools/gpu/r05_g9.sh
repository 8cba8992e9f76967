cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base occ4 occ2 basic4 > $O/ab_occ_principled.log 2>&1; cat $O/ab_occ_principled.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base occ4 occ2 basic4 > $O/ab_occ_standin.log 2>&1; cat $O/ab_occ_standin.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail or wide" -n 4 2>&1 | tail -2
