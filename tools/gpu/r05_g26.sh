cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x; mkdir -p $O
bash tools/ab_env.sh 20 "IGH_NODE_QUANT=1" "IGH_NODE_QUANT=1 IGD_NODE_FORMAT=full" "-" > $O/ab_q8_headline.log 2>&1; cat $O/ab_q8_headline.log
