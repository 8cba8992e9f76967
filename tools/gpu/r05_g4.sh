cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quantised or standin" -n 4 > $O/pytest_q8.log 2>&1
echo "pytest rc=$?" >> $O/pytest_q8.log
tail -5 $O/pytest_q8.log
bash tools/run_fetch_calibration.sh r05c > $O/fetch_cal.log 2>&1
cat $O/r05c_fetch_calibration.txt
# stand-in 16 M (lean mix: the traversal workload): Node8 records vs quantised records
bash tools/standin_quick.sh r05c 16000000 16 "IGD_NODE_FORMAT=full" "-" "IGD_NODE_FORMAT=full" "-" > $O/standin16M.log 2>&1
cat $O/standin16M.log
IGH_NODE_QUANT=0 bash tools/standin_quick.sh r05c 16000000 16 "-" > $O/standin16M_noquant.log 2>&1
cat $O/standin16M_noquant.log
