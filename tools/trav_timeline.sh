# tools/trav_timeline.sh: per-wave timeline summaries of the traversal launches (variant build: ONLY=traverse tools/build_variant.sh ttime -DIG_TRAV_TIMELINE)
#  1. igd_traverse on random rays at growing launch sizes   2. the launches of one 20-iteration wavefront rendered as rank 0 of 8
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/roundtrace; mkdir -p $OUT
export IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_ttime.so
rm -f $OUT/timeline_sizes.txt $OUT/timeline_rankof8.txt
IGD_TRAV_TIMELINE=$OUT/timeline_sizes.txt python tools/traverse_sizes.py 23 > $OUT/timeline_sizes_times.txt 2>&1
IGD_TRAV_TIMELINE=$OUT/timeline_rankof8.txt python bench.py --steps 20 --warmup 0 --as-rank-of 8 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic --no-stage-timers > /dev/null 2>&1
