cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/g58.log 2>&1; tail -3 gpurun_out/g58.log
