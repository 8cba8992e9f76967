cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
bash tools/ab_env.sh 20 "-" "IGD_LEAN_SORT=1" 2>&1 | tee $O/ab_lean_sort.txt
IGD_LEAN_SORT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rounds and (radiance or reproducible or row_sharding or full_size)" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
