cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "brick or expression" > gpurun_out/g52.log 2>&1; tail -3 gpurun_out/g52.log
