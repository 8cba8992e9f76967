cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
export STANDIN_MATERIALS=divergent
bash tools/standin_quick.sh r05e 1000000 16 "IGH_NODE_QUANT=0" "IGH_NODE_QUANT=1" "IGH_NODE_QUANT=0" "IGH_NODE_QUANT=1" > $O/standin1M_quant.log 2>&1; cat $O/standin1M_quant.log
export STANDIN_MATERIALS=lean
bash tools/standin_quick.sh r05e 16000000 16 "IGD_NODE_REPEAT=0" "IGD_NODE_REPEAT=3" "IGD_NODE_REPEAT=6" > $O/standin16M_repeat.log 2>&1; cat $O/standin16M_repeat.log
