cd $GRAFT_REPO_ROOT
TRACE_NAME=whole_timed bash tools/round_trace.sh
TRACE_NAME=rankof8_timed BENCH_ARGS="--as-rank-of 8" bash tools/round_trace.sh
TRACE_NAME=principled_timed BENCH_ARGS="--scene scenes/diamond_scene_principled.json" bash tools/round_trace.sh
tail -22 gpurun_out/roundtrace/principled_timed.txt
