#!/bin/bash
# Runs ON the GPU box (via gpurun): rocprofv3 kernel statistics of the default bench command, then the PMC
# passes (one counter group per run, combined only with --kernel-trace as the pool requires).
# usage: tools/collect_profiles.sh <tag>        outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
# 1. the bench line on its own (no profiler attached)
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# 2. same command under --kernel-trace --stats
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python bench.py > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
# 3. PMC passes (shorter run of the same workload: counters serialise the kernels)
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$C" -o pmc -- python bench.py --no-cpu-baseline > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace -d "$OUT/pmc_SQ" -o pmc -- python bench.py --no-cpu-baseline > "$OUT/pmc_SQ.json" 2> "$OUT/pmc_SQ.err"
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace -d "$OUT/pmc_MEM" -o pmc -- python bench.py --no-cpu-baseline > "$OUT/pmc_MEM.json" 2> "$OUT/pmc_MEM.err"
find "$OUT" -name "*.db" -size +30M -delete
ls -la "$OUT" "$OUT"/*/ 2>/dev/null | head -40
tail -c 600 "$OUT/bench.json"
