cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_single_rank or rccl or comm" > gpurun_out/g47.log 2>&1; grep -E "passed|failed|error" gpurun_out/g47.log | tail -3
