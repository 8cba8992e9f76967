cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ap; mkdir -p $O
bash tools/ab.sh 20 base ria16 ria32 ri16 ri20 > $O/ab_refill_headline.log 2>&1; cat $O/ab_refill_headline.log
bash tools/ab.sh 20 base ria16 ria32 ri16 ri20 > $O/ab_refill_headline2.log 2>&1; cat $O/ab_refill_headline2.log
