cd $GRAFT_REPO_ROOT
python tools/traverse_sizes.py 25 2>&1 | tee gpurun_out/roundtrace/traverse_sizes.txt
