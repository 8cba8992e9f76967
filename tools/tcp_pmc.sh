#!/bin/bash
# Runs ON the GPU box: how busy the vector L1 path (TA / TCP) is under the traversal kernels. One rocprofv3 --pmc pass per counter group
# (with --kernel-trace only, as the pool requires) of a short bench run; prints per-kernel sums per launch.
# usage: tools/tcp_pmc.sh <tag> [bench args ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
i=0
for G in "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G --kernel-trace -d "$OUT/g$i" -o pmc -- python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic "$@" > /dev/null 2> "$OUT/g$i.err" || tail -3 "$OUT/g$i.err"
  db=$(find "$OUT/g$i" -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py pmc "$db" 2>/dev/null | grep "k_traverse<false, false, false, false>\|k_traverse<true, false, false, false>\|k_shade<false\|^kernel"
done
find "$OUT" -name "*.db" -delete
