#!/bin/bash
# Runs ON the GPU box: the VALU issue calibration (tools/valu_calibration.hip), once on its own and once per PMC group
# (counters only with --kernel-trace, as the pool requires). Output: gpurun_out/<tag>/<tag>_valu_calibration.txt
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
BIN=$ROOT/ignis_amd/lib/valu_calibration
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/valu_calibration.hip -o "$BIN"
F=$OUT/${TAG}_valu_calibration.txt
{
  echo "## tools/valu_calibration.hip on $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//')"
  timeout 60 "$BIN"
} > "$F" 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/cal_pmc" -o pmc -- "$BIN" > "$OUT/cal_pmc.out" 2> "$OUT/cal_pmc.err"
python - "$OUT" >> "$F" <<'PY'
import sqlite3, sys, glob
dbs = glob.glob(sys.argv[1] + "/cal_pmc/**/*.db", recursive=True)
if not dbs:
    print("## no PMC database"); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
rows = {}
# launches of one kernel come in the order w = 1,1,1,2,2,2,3,3,3,4,4,4 (three repetitions each)
q = "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection order by dispatch_id"
for k, d, c, v, dur in cur.execute(q):
    rows.setdefault((k, d), {"ns": dur})[c] = v
print("\n## PMC per launch (third repetition of each configuration): SQ_* counters are summed over the chip")
print(f"{'kernel':28s} {'w/SIMD':>6s} {'INSTS_VALU':>12s} {'ACTIVE_INST_VALU':>17s} {'ACT/INST':>9s} {'BUSY_CYCLES':>12s} {'WAVE_CYCLES':>12s} {'GUI_ACTIVE':>11s} {'us':>8s}")
per = {}
for (k, d), r in sorted(rows.items(), key=lambda x: x[0][1]):
    per.setdefault(k, []).append(r)
for k, lst in per.items():
    name = k.split("<")[1].split(">")[0] if "<" in k else k
    for i, r in enumerate(lst):
        if i % 3 != 2:
            continue
        w = i // 3 + 1
        iv, av = r.get("SQ_INSTS_VALU", 0), r.get("SQ_ACTIVE_INST_VALU", 0)
        print(f"k_stream<{name:>3s}>{'':14s} {w:6d} {iv:12.0f} {av:17.0f} {av / iv if iv else 0:9.3f} {r.get('SQ_BUSY_CYCLES', 0):12.0f} {r.get('SQ_WAVE_CYCLES', 0):12.0f} {r.get('GRBM_GUI_ACTIVE', 0):11.0f} {r['ns'] / 1e3:8.1f}")
PY
find "$OUT/cal_pmc" -name "*.db" -delete
cat "$F"
