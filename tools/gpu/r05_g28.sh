cd $GRAFT_REPO_ROOT
O=gpurun_out/r05aa; mkdir -p $O
bash tools/ab.sh 20 base defer > $O/ab_defer_headline.log 2>&1; cat $O/ab_defer_headline.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base defer > $O/ab_defer_standin.log 2>&1; cat $O/ab_defer_standin.log
IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_defer.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 2>&1 | tail -2
