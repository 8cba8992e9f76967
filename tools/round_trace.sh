# tools/round_trace.sh: the kernel timeline of the TIMED render of `bench.py --steps 20 --warmup 5 $BENCH_ARGS` (rocprofv3 kernel trace; the render
# after it is bench.py's replay with the work counters on):
# every launch with its start, duration and the gap to the previous launch's end, to see what a small wavefront's rounds are made of.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/roundtrace; mkdir -p $OUT
NAME=${TRACE_NAME:-whole}
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic $BENCH_ARGS > /dev/null 2> $OUT/err.txt
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/$NAME.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gens = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"]]
rows = rows[gens[-2]:gens[-1]]
t0 = int(rows[0]["Start_Timestamp"])
def short(n):
    m = re.match(r"(?:void )?(?:igdev::)?(k_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]
prev_end = None
tot = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%10.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev_end = max(prev_end or 0, e)
    k = tot.setdefault(n, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
print("# span %.1f us" % ((prev_end - t0) / 1e3))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("# %-50s %4d launches %10.1f us" % (n, c, t))
PY
find $OUT/t -type f -delete
