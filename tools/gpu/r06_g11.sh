cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
bash tools/ab.sh 20 base fastrcp 2>&1 | tee $O/ab_fastrcp.txt
IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_fastrcp.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "radiance or analytic or principled or selectors" > $O/pytest_fastrcp.log 2>&1; tail -3 $O/pytest_fastrcp.log
bash tools/gpu/r06_final.sh
