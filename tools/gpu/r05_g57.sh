cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/g57.log 2>&1; tail -3 gpurun_out/g57.log
bash tools/ab.sh 20 base 2>&1 | tail -2
