cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
bash tools/ab.sh 20 base prefetch > $O/ab_prefetch_headline.log 2>&1; cat $O/ab_prefetch_headline.log
bash tools/ab_scene.sh scenes/many_point_lights.json 32 base prefetch > $O/ab_prefetch_mpl.log 2>&1; cat $O/ab_prefetch_mpl.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base prefetch > $O/ab_prefetch_standin.log 2>&1; cat $O/ab_prefetch_standin.log
IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_prefetch.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 2>&1 | tail -2
