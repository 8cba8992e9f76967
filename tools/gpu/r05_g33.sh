cd $GRAFT_REPO_ROOT
TRACE_NAME=whole bash tools/round_trace.sh
TRACE_NAME=rankof8 BENCH_ARGS="--as-rank-of 8" bash tools/round_trace.sh
tail -25 gpurun_out/roundtrace/rankof8.txt
