cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
bash tools/ab.sh 20 base nosortidx 2>&1 | tee $O/ab_sortidx.txt
bash tools/ab.sh 256 base nosortidx 2>&1 | tee -a $O/ab_sortidx.txt
