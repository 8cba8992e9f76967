"""Summarise rocprofv3 (rocpd sqlite) outputs into small text / JSON files for profiles/.

usage:
  python tools/prof_summary.py stats <stats.db>                      kernel table of `--kernel-trace --stats`
  python tools/prof_summary.py pmc <pmc.db> [<pmc.db> ...]           per-kernel sum / per-launch mean of each counter
  python tools/prof_summary.py traffic <fetch.db> <write.db> [<sq.db>|-] [steps] [bench.json]   JSON: HBM-side bytes per launch per kernel (+ VALU lane utilisation)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB. Calibration inside the same run (known byte counts):
k_generate writes exactly 68 B per camera ray (WRITE_SIZE matches to 6 digits -> no correction); k_copy_paths reads
what it writes (68 B per path, 16 B/lane coalesced loads) and FETCH_SIZE shows exactly half of its WRITE_SIZE, i.e.
the x2 correction MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 applies to our stream reads.
"""
import json
import os
import sqlite3
import sys


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    print(f"# kernel stats: {path}")
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:72]:72s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
    print()


def counters(path):
    cur = sqlite3.connect(path).cursor()
    out = {}
    q = "select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
    for k, c, n, s, a, d in cur.execute(q):
        out.setdefault(k, {})[c] = {"launches": n, "sum": s, "mean": a, "avg_ns": d}
    return out


def pmc(path):
    print(f"# counters: {path}")
    print(f"{'kernel':60s} {'counter':30s} {'launches':>8s} {'sum':>16s} {'mean/launch':>14s} {'avg_ns':>10s}")
    for k, cs in sorted(counters(path).items()):
        for c, v in sorted(cs.items()):
            print(f"{k[:60]:60s} {c:30s} {v['launches']:8d} {v['sum']:16.1f} {v['mean']:14.2f} {v['avg_ns']:10.0f}")
    print()


def short(name):
    n = name.replace("void ", "").replace("igdev::", "")
    return n.split("(")[0]


def traffic(fetch_db, write_db, sq_db=None, steps=None, bench_json=None):
    f, w = counters(fetch_db), counters(write_db)
    sq = counters(sq_db) if sq_db else {}
    res = {"unit": "bytes per launch", "fetch_correction": 2.0, "steps": steps,
           "note": "FETCH_SIZE/WRITE_SIZE in KiB from separate --pmc passes; reads doubled (gfx950 coalesced-read "
                   "correction, confirmed in-run by k_copy_paths: FETCH == WRITE / 2 for a pure copy), writes as reported "
                   "(k_generate: 68 B/ray exactly)",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fr = f.get(k, {}).get("FETCH_SIZE")
        wr = w.get(k, {}).get("WRITE_SIZE")
        if not fr or not wr:
            continue
        res["kernels"][short(k)] = {
            "launches": fr["launches"],
            "fetch_kib_raw": round(fr["mean"], 2),
            "write_kib_raw": round(wr["mean"], 2),
            "hbm_bytes": int((2.0 * fr["mean"] + wr["mean"]) * 1024),
        }
        q = sq.get(k)
        if q and all(c in q and q[c]["sum"] > 0 for c in ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES")):
            # lane utilisation = active lanes per issued VALU instruction / 64; wait share = waves parked on s_waitcnt
            res["kernels"][short(k)]["valu_lane_utilisation"] = round(q["SQ_THREAD_CYCLES_VALU"]["sum"] / (64.0 * q["SQ_ACTIVE_INST_VALU"]["sum"]), 4)
            res["kernels"][short(k)]["wave_wait_share"] = round(q["SQ_WAIT_ANY"]["sum"] / q["SQ_WAVE_CYCLES"]["sum"], 4)
            res["kernels"][short(k)]["wave_issue_share"] = round(q["SQ_ACTIVE_INST_ANY"]["sum"] / q["SQ_WAVE_CYCLES"]["sum"], 4) if "SQ_ACTIVE_INST_ANY" in q else None
            if "SQ_INSTS_VALU" in q:
                # VALU line: wave-instructions issued x 64 lanes = issue slots used; x lane utilisation = lane operations that did work.
                # Peak: 256 CUs x 4 SIMDs x 32 lanes per cycle (a wave64 v_fma_f32 takes 2 cycles, MI355X_MICROARCH.md) x 2.4 GHz.
                lanes = q["SQ_INSTS_VALU"]["mean"] * 64.0
                cycles_per_inst = 2.0
                if bench_json:
                    for tag in (os.path.basename(bench_json).split("_")[0], "r05"):  # (this round's accounting, else the last one collected)
                        acc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag + "_issue_accounting.json")
                        if os.path.exists(acc):
                            cycles_per_inst = json.load(open(acc))["valu_cycles_per_inst"]
                            break
                util = res["kernels"][short(k)]["valu_lane_utilisation"]
                secs = q["SQ_INSTS_VALU"]["avg_ns"] * 1e-9
                peak = 256 * 4 * 32 * 2.4e9
                res["kernels"][short(k)]["valu_insts_per_launch"] = int(q["SQ_INSTS_VALU"]["mean"])
                res["kernels"][short(k)]["valu_lane_ops_per_launch"] = int(lanes * util)
                # issue share: instructions x the mix's cycles per instruction (profiles/<tag>_issue_accounting.json, tools/issue_accounting.py: the
                # closest-hit traversal kernel's dynamic opcode histogram x calibrated prices) against 1024 SIMDs at the clock under load —
                # the same formula and the same price bench.py uses. Without that file: 2 cycles per instruction (the guide's v_fma_f32).
                res["kernels"][short(k)]["valu_issue_frac"] = round(q["SQ_INSTS_VALU"]["mean"] * cycles_per_inst / (1024 * 2.1e9 * secs), 4) if secs > 0 else None
                res["kernels"][short(k)]["valu_issue_frac_cycles_per_inst"] = cycles_per_inst
                res["kernels"][short(k)]["avg_ns_under_pmc"] = int(q["SQ_INSTS_VALU"]["avg_ns"])
    # per-ray figures of the closest-hit traversal kernel, so that bench.py can scale them to whatever step count it is run with
    # (the driver's --steps differs from the profiled one): rays per launch of the profiled command from its own bench line
    if bench_json:
        try:
            b = json.loads(open(bench_json).read().strip().split("\n")[-1])
            rays = b["rays"]["camera"] + b["rays"]["bounce"]
            launches = b["roofline"]["launches"]
            tk = next((v for k, v in res["kernels"].items() if k.startswith("k_traverse<false, false, false")), None)
            if tk and launches and rays:
                rpl = rays / launches
                res["closest_hit_per_ray"] = {"rays_per_launch_profiled": rpl, "hbm_bytes": tk["hbm_bytes"] / rpl,
                                              "valu_insts": tk.get("valu_insts_per_launch", 0) / rpl,
                                              "valu_lane_ops": tk.get("valu_lane_ops_per_launch", 0) / rpl}
        except Exception as e:  # noqa: BLE001
            res["closest_hit_per_ray_error"] = str(e)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        kernel_stats(sys.argv[2])
    elif mode == "pmc":
        for p in sys.argv[2:]:
            pmc(p)
    elif mode == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None, int(sys.argv[5]) if len(sys.argv) > 5 else None,
                sys.argv[6] if len(sys.argv) > 6 else None)
    else:
        sys.exit(__doc__)
