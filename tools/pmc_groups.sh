#!/bin/bash
# Runs ON the GPU box: extra PMC groups of the bench command (one rocprofv3 run per group, counters only with --kernel-trace).
# usage: [PMC_ONLY="0 4"] tools/pmc_groups.sh <tag> <label> [bench args ...]      groups: see PMCG below (PMC_ONLY: indices to run)
#   output: gpurun_out/<tag>/<tag>_pmc_<label>.txt  (tools/prof_summary.py pmc: per-kernel sum / mean per launch of each counter)
TAG=$1; LABEL=$2; shift; shift
ARGS=${*:---steps 32 --warmup 32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
PMCG=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
 "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_WAIT_INST_LDS SQ_WAVES"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
)
F=$OUT/${TAG}_pmc_$LABEL.txt
: > "$F"
i=0
for G in "${PMCG[@]}"; do
  if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then i=$((i+1)); continue; fi
  D=$OUT/pmcg_${LABEL}_$i
  timeout 600 rocprofv3 --pmc $G --kernel-trace -d "$D" -o pmc -- python bench.py $ARGS --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic > "$D.json" 2> "$D.err"
  DB=$(find "$D" -name "*.db" | head -1)
  if [ -n "$DB" ]; then python tools/prof_summary.py pmc "$DB" | grep -v "rocclr\|k_round_end\|k_secondary_end\|k_copy" >> "$F"; else echo "## group $i failed: $G" >> "$F"; tail -3 "$D.err" >> "$F"; fi
  rm -rf "$D"
  i=$((i+1))
done
cat "$F" | cut -c1-150
