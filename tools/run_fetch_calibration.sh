#!/bin/bash
# Runs ON the GPU box: tools/fetch_calibration.hip plain (times) and under rocprofv3 --pmc FETCH_SIZE / the L2's request counters
# (separate passes, --kernel-trace only), summary -> gpurun_out/<tag>/<tag>_fetch_calibration.txt (copied to profiles/ by hand).
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BIN=$ROOT/ignis_amd/lib/fetch_calibration
[ -x "$BIN" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 "$ROOT/tools/fetch_calibration.hip" -o "$BIN"
R=$OUT/${TAG}_fetch_calibration.txt
{
echo "# tools/fetch_calibration.hip on $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//')"
echo "## plain run"
"$BIN"
for C in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_HIT_sum" "TCC_REQ_sum TCC_READ_sum"; do
  D=$OUT/fc_$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d "$D" -o pmc -- "$BIN" > /dev/null 2> "$D.err"
  echo "## rocprofv3 --pmc $C   (per kernel launch, in launch order)"
  python3 - "$D" <<'PY'
import glob, sqlite3, sys
dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
if not dbs:
    print("  (no database: counter not available?)"); sys.exit(0)
db = sqlite3.connect(dbs[0])
# (the view tools/prof_summary.py reads: one row per kernel launch and counter; k_node<16, 14> runs twice: cold, then over the warm window)
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
order = next((c for c in ("dispatch_id", "start", "start_timestamp", "id") if c in cols), None)
q = "select kernel_name, counter_name, value, duration from counters_collection" + (f" order by {order}" if order else "")
for kn, cn, v, d in db.execute(q):
    print(f"  {kn[:64]:64s} {cn:28s} {v:16.0f}   {d / 1e6:9.3f} ms")
PY
done
} > "$R" 2>&1
find "$OUT" -name "*.db" -delete
cat "$R"
