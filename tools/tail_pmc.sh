#!/bin/bash
# Runs ON the GPU box: per-dispatch counters of the k_tail passes of the last (un-overlapped) tail of a `--steps 20 --warmup 5` run:
# waves that did something, instructions and cycles per wave — what a late pass's ~100 us per bounce is made of.
# usage: tools/tail_pmc.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tailpmc; mkdir -p $OUT
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  rm -rf $OUT/t
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/t -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic "$@" > /dev/null 2> $OUT/err.txt
  f=$(find $OUT/t -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if "k_tail" not in r["Kernel_Name"]:
        continue
    d = by.setdefault(int(r["Dispatch_Id"]), {})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
last = list(by.items())[-11:]
names = sorted({k for _, d in last for k in d})
print("dispatch " + " ".join("%16s" % n for n in names))
for i, d in last:
    print("%8d " % i + " ".join("%16.0f" % d.get(n, 0) for n in names))
PY
done
rm -rf $OUT/t
