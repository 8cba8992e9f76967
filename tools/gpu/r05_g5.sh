cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
bash tools/run_fetch_calibration.sh r05c > $O/fetch_cal.log 2>&1
cat $O/r05c_fetch_calibration.txt
bash tools/run_standin.sh r05 16000000 16 lean _standin > gpurun_out/r05_collect_standin.log 2>&1
tail -3 gpurun_out/r05_collect_standin.log
