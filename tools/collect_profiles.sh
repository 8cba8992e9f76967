#!/bin/bash
# Runs ON the GPU box (via gpurun): the bench line, rocprofv3 kernel statistics of the same command, then the PMC
# passes (one counter group per run, combined only with --kernel-trace as the pool requires), and the summaries
# that go into profiles/ (tools/prof_summary.py).
# usage: tools/collect_profiles.sh <tag> [suffix] [bench args ...]
#   outputs under gpurun_out/<tag>/:  <tag>_bench<suffix>.json  <tag>_rocprofv3_stats<suffix>.txt  <tag>_rocprofv3_pmc<suffix>.txt  <tag>_traffic<suffix>.json
set -u
TAG=${1:-r03}
SUF=${2:-}
shift; shift
ARGS="$*"
# the committed counter summary bench.py falls back to is named by the caller (two stand-ins are both standin.json)
[ -n "$SUF" ] && ARGS="$ARGS --profile-key ${SUF#_}"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
STEPS=$(python - "$@" <<'EOF'
import sys
a = sys.argv[1:]
print(a[a.index("--steps") + 1] if "--steps" in a else 256)
EOF
)
# 1. the bench line on its own (no profiler attached)
timeout 900 python bench.py $ARGS > "$OUT/${TAG}_bench$SUF.json" 2> "$OUT/bench$SUF.err"
# 2. same command under --kernel-trace --stats
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats$SUF" -o stats -- python bench.py $ARGS --no-cpu-baseline --no-live-traffic --no-extra-configs > "$OUT/bench_under_rocprof$SUF.json" 2> "$OUT/stats$SUF.err"
# 3. PMC passes of the same command (same step count: the per-launch averages then refer to the same launches)
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$C$SUF" -o pmc -- python bench.py $ARGS --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic > "$OUT/pmc_$C$SUF.json" 2> "$OUT/pmc_$C$SUF.err"
done
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d "$OUT/pmc_SQ$SUF" -o pmc -- python bench.py $ARGS --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic > "$OUT/pmc_SQ$SUF.json" 2> "$OUT/pmc_SQ$SUF.err"
timeout 900 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace -d "$OUT/pmc_MEM$SUF" -o pmc -- python bench.py $ARGS --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic > "$OUT/pmc_MEM$SUF.json" 2> "$OUT/pmc_MEM$SUF.err"
db() { find "$OUT/$1" -name "*.db" | head -1; }
python tools/prof_summary.py stats "$(db stats$SUF)" > "$OUT/${TAG}_rocprofv3_stats$SUF.txt" 2>> "$OUT/summary.err"
python tools/prof_summary.py pmc "$(db pmc_FETCH_SIZE$SUF)" "$(db pmc_WRITE_SIZE$SUF)" "$(db pmc_SQ$SUF)" "$(db pmc_MEM$SUF)" > "$OUT/${TAG}_rocprofv3_pmc$SUF.txt" 2>> "$OUT/summary.err"
python tools/prof_summary.py traffic "$(db pmc_FETCH_SIZE$SUF)" "$(db pmc_WRITE_SIZE$SUF)" "$(db pmc_SQ$SUF)" "$STEPS" "$OUT/${TAG}_bench$SUF.json" > "$OUT/${TAG}_traffic$SUF.json" 2>> "$OUT/summary.err"
if [ -z "$SUF" ]; then
  # the issue accounting of the closest-hit traversal kernel (tools/issue_accounting.py): event counts of the profile build, the calibration,
  # then the traffic summary once more so that its valu_issue_frac is priced with the derived cycles per instruction
  V=$ROOT/ignis_amd/lib/var
  if [ -f "$V/libig_device_hip_tprof1.so" ]; then
    mkdir -p "$ROOT/profiles"
    IGD_LIBRARY=$V/libig_device_hip_tprof1.so python tools/trav_events.py > "$OUT/${TAG}_trav_events_closest.json" 2>> "$OUT/summary.err"
    IGD_LIBRARY=$V/libig_device_hip_tprof2.so python tools/trav_events.py > "$OUT/${TAG}_trav_events_any.json" 2>> "$OUT/summary.err"
    [ -f "$ROOT/profiles/${TAG}_valu_calibration.txt" ] || timeout 240 bash tools/run_valu_calibration.sh "$TAG" > /dev/null 2>&1
    for f in trav_events_closest.json traffic.json rocprofv3_pmc.txt valu_calibration.txt; do [ -f "$OUT/${TAG}_$f" ] && cp "$OUT/${TAG}_$f" "$ROOT/profiles/${TAG}_$f"; done
    python tools/issue_accounting.py "$TAG" > /dev/null 2>> "$OUT/summary.err" && cp "$ROOT/profiles/${TAG}_issue_accounting.json" "$OUT/"
    python tools/prof_summary.py traffic "$(db pmc_FETCH_SIZE$SUF)" "$(db pmc_WRITE_SIZE$SUF)" "$(db pmc_SQ$SUF)" "$STEPS" "$OUT/${TAG}_bench$SUF.json" > "$OUT/${TAG}_traffic$SUF.json" 2>> "$OUT/summary.err"
  fi
fi
find "$OUT" -name "*.db" -delete
ls -la "$OUT" | head -40
tail -c 400 "$OUT/${TAG}_bench$SUF.json"
