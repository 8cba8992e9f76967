cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule or reproducible or row_sharding" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
STANDIN_MATERIALS=divergent bash tools/standin_quick.sh r06a 1000000 8 "-" "IGD_RAY_SORT=0" "IGD_RAY_SORT=1 IGD_RAY_SORT_STREAMS=1" "IGD_RAY_SORT=1 IGD_RAY_SORT_STREAMS=2" "IGD_RAY_SORT=1" "IGD_RAY_SORT=1 IGD_RAY_SORT_ORDER=cell" "IGD_RAY_SORT=1 IGD_RAY_SORT_BITS=5" "IGD_RAY_SORT=1 IGD_RAY_SORT_BITS=9" 2>&1 | tee $O/ab.txt
