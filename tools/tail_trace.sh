cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tailtrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tailtrace/t -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic $BENCH_ARGS > /dev/null 2> gpurun_out/tailtrace/err.txt
f=$(find gpurun_out/tailtrace/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
tail = [r for r in rows if "k_tail" in r["Kernel_Name"]]
last = tail[-11:]
prev_end = None
allk = rows
# end of last non-tail, non-resolve kernel before final tail
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("k_tail  dur %8.1f us  gap_before %7.1f us  grid %s" % ((e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev_end = e
print("final tail span %.1f us" % ((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))
PY
find gpurun_out/tailtrace/t -type f -delete
