cd $GRAFT_REPO_ROOT
O=gpurun_out/r05am; mkdir -p $O
bash tools/ab.sh 20 base pre2o3 pre2o4 pre1o3 > $O/ab_prefetch2_headline.log 2>&1; cat $O/ab_prefetch2_headline.log
