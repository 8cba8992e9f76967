#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric for the Ignis hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one render iteration (Runtime::step, src/runtime/Runtime.cpp:334-387) of the workload
BASELINE.json's metric is quoted on: scenes/diamond_scene.json, 1920x1080, path integrator,
spi 8 (64 spp = 8 steps). Inputs (scene tables) are resident in HBM before the timed region. The default K (256
steps = 2048 spp, 8 wavefronts of 32 iterations) keeps the timed region above 3 s; the literal 64-spp configuration (8 steps from an idle device) is
timed separately and printed as `literal_config`.

N > 1 (one process per GPU): the film is tile-sharded — rank r renders film rows r, r + N, ... of every iteration
(igd_render_settings.row_offset / row_stride; the device batches a rank's small iterations into full-size wavefronts)
with no data-path exchange; the only collective, inside the timed region, is ONE gather of the owned rows to rank 0
over RCCL (W x H x 12 / N bytes per rank; ignis_amd.sharding.gather_rows). K iterations in total whatever N is:
"scaling": "strong". N = 1 runs the same code path without a process group. `--sharding iterations` keeps the other
partition (rank r renders K whole-film iterations r*K .. r*K+K-1, reduce(SUM); "weak").

`roofline.traffic` is measured in the run itself at N = 1 (two more passes of the same command under `rocprofv3 --pmc FETCH_SIZE` /
`WRITE_SIZE --kernel-trace` as child processes, ~10 s each; `--no-live-traffic` or a failing profiler fall back to the figure
profiles/<tag>_traffic.json holds, which `traffic_from_profiles` always shows next to it).

Prints ONE JSON line (rank 0): Mrays/s = (camera + bounce + shadow rays) / s as the reference counts them
(src/runtime/Statistics.cpp:286-290), plus Msamples/s (src/frontend/cli/main.cpp:134), the roofline block of the
dominant kernel (closest-hit traversal) and the CPU baseline (oracle) on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPI, SEED = 1920, 1080, 8, 1
SCENE = os.path.join(ROOT, "scenes", "diamond_scene.json")
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU peak for the "valu" line: 256 CUs x 4 SIMDs x 32 lanes/cycle (a wave64 v_fma_f32 takes 2 cycles, MI355X_MICROARCH.md) x 2.4 GHz
VALU_PEAK_GLANE_OPS = 256 * 4 * 32 * 2.4
SHADER_GHZ = 2.1            # effective shader clock of the traversal launches (GRBM_GUI_ACTIVE / duration; 2.4 GHz is the boost limit)
PROFILE_TAG = "r06"  # profiles/<tag>_traffic[_<profile key>].json: PMC summary of this command, tools/collect_profiles.sh
# cycles a wave64 VALU instruction of the shading kernels occupies on average (the lean kernel's mix, 3.03; the traversal kernels'
# price comes from profiles/<tag>_issue_accounting.json)
SHADE_CYCLES_PER_INST = 3.0
SQ_GROUP = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"]
# which kernels make up a stage of a bounce round (short names as rocprofv3 reports them, `void igdev::` stripped)
# (the traversal kernels of the timed steps: <ANY_HIT, STATS = false, DEEP = false, ...>; the counter replay's STATS instantiations and the — empty — DEEP launches are not them)
STAGE_KERNELS = {"k_traverse<closest>": ("k_traverse<false, false, false",), "k_traverse<any>": ("k_traverse<true, false, false",), "k_shade": ("k_shade<", "k_bin_", "k_round_end")}


def valu_cycles_per_inst():
    """SIMD cycles one wave64 VALU instruction of k_traverse<closest> occupies on average: the kernel's dynamic opcode histogram (static
    counts of its parts x how often each runs, scaled to the measured SQ_INSTS_VALU) x the calibrated price of each opcode class —
    profiles/<tag>_issue_accounting.json, written by tools/issue_accounting.py from profiles/<tag>_valu_calibration.txt,
    _trav_events_closest.json and _traffic.json. None when that file is missing (the `valu.issue_frac` field is then left out)."""
    for tag in (PROFILE_TAG, "r05"):
        p = os.path.join(ROOT, "profiles", f"{tag}_issue_accounting.json")
        if os.path.exists(p):
            return json.load(open(p))["valu_cycles_per_inst"]
    return None
DEFAULT_STEPS = 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=DEFAULT_STEPS)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-literal-config", action="store_true", help="skip the separate 8-step (64 spp) timing")
    ap.add_argument("--no-stage-timers", action="store_true", help="experiments only: no HIP-event stage timers in the timed loop (roofline.achieved becomes 0)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spi", type=int, default=SPI)
    ap.add_argument("--sharding", choices=("rows", "iterations"), default="rows", help="N > 1: how camera samples are split")
    ap.add_argument("--collective", choices=("gather", "reduce"), default="gather", help="rows sharding: gather of the owned rows (default) or reduce(SUM) of whole framebuffers")
    ap.add_argument("--as-rank-of", type=int, default=0, help="experiments only: one process renders what rank 0 of N row-sharding ranks would (estimate of per-GPU throughput at N GPUs)")
    ap.add_argument("--scene", default=SCENE, help="other scene file (not the headline workload), e.g. tools/make_standin_scene.py output")
    ap.add_argument("--dist", choices=("rccl", "torch"), default="rccl",
                    help="N > 1: rccl = the device library's own RCCL communicator (igd_comm_*, ignis_amd/comm.py; no torch in the process), "
                         "torch = torch.distributed with the nccl backend (the path of rounds 1 - 4)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes of the same command as child processes); "
                         "the figure then comes from profiles/<tag>_traffic.json")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the `configs` array (BASELINE configs 3 and 4 measured next to the headline)")
    ap.add_argument("--profile-key", default=None,
                    help="names this workload's committed counter summary, profiles/<tag>_traffic_<key>.json (the fallback of the live measurement); "
                         "the headline scene needs none, another scene without a key has no fallback (two scene files may share a basename)")
    return ap.parse_args()


def effective_cpus():
    """CPUs this process may really use: the affinity mask capped by the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) — the GPU
    boxes show 256 hardware threads under a quota of 16 CPUs, where 256 worker threads only add switching."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return (max(1, min(n, int(quota + 0.5))) if quota else max(1, n)), n, quota


def stream_bytes(n_rays):
    """Bytes of the ray streams a closest-hit launch has to move whatever the caches do: 40 B read
    (id, org, dir, tmin, tmax, flags) + 20 B hit written per ray."""
    return 60 * n_rays


def geometry_bytes(nodes, tris, leaves, node_bytes=256):
    """SURVEY.md 8(d) geometry term with this backend's layouts: 256 B per Node8 fetched (128 B when the scene runs on the quantised
    node records, igd_node_bytes) + 52 B per triangle tested (a 208 B Tri4 packet holds 4) + 96 B per EntityLeaf1 tested."""
    return node_bytes * nodes + 52 * tris + 96 * leaves


def _primbvh_nodes(scene):
    """Node8 records of all shape BVHs: the "trimesh_primbvh" fix table is {u32 nodes, u32 packets, pad, pad} Node8[] Tri4[] per shape."""
    import ctypes as C
    s = scene.scene
    seen, total = set(), 0
    for i in range(s.scene_leaf_count):
        off = (((int(s.scene_leaves[i].user[1]) & 0xFFFFFFFF) << 32) | (int(s.scene_leaves[i].user[0]) & 0xFFFFFFFF)) * 4
        if off in seen or off + 16 > s.primbvh_size:
            continue
        seen.add(off)
        total += int(C.cast(C.c_void_p(C.addressof(s.primbvh.contents) + off), C.POINTER(C.c_uint32))[0])
    return total


def _short(name):
    return name.replace("void ", "").replace("igdev::", "").split("(")[0]


def pmc_passes(scene, W, H, spi, steps, groups):
    """This workload under the counters, in THIS run: the same command once per counter group as a child
    process under `rocprofv3 --pmc <group> --kernel-trace` (separate passes, counters alone with the kernel trace, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes). Returns ({kernel: {counter: {"sum", "launches", "ns"}}}, the child's bench line)
    or (None, the reason) when rocprofv3 is not there or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    if pmc_passes.broken:  # (a profiler that failed or timed out once is not asked again for the next workload: the line must come out in minutes)
        return None, pmc_passes.broken
    ctr, line = {}, None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    try:
        for gi, group in enumerate(groups):
            out = os.path.join(tmp, f"g{gi}")
            cmd = [exe, "--pmc"] + list(group) + ["--kernel-trace", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", str(steps), "--warmup", str(steps), "--width", str(W), "--height", str(H), "--spi", str(spi),
                   "--scene", os.path.abspath(scene), "--no-cpu-baseline", "--no-literal-config", "--no-extra-configs", "--no-live-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                pmc_passes.broken = f"rocprofv3 --pmc {' '.join(group)} failed (rc {r.returncode})"
                return None, pmc_passes.broken
            cur = sqlite3.connect(dbs[0]).cursor()
            q = "select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection group by kernel_name, counter_name"
            for k, c, n, v, d in cur.execute(q):
                ctr.setdefault(_short(k), {})[c] = {"sum": float(v), "launches": int(n), "ns": float(d or 0)}
            line = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # (a profiler hiccup must not cost the bench line)
        pmc_passes.broken = f"{type(e).__name__}: {e}"
        return None, pmc_passes.broken
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return ctr, line


pmc_passes.broken = None


def stage_evidence(ctr, child_line, stage, ms_launch, rays_scale=1.0):
    """What the counters say about one stage of a bounce round (STAGE_KERNELS): HBM-side bytes per launch (2 x FETCH_SIZE — the guide's
    gfx950 correction, which profiles/r05_fetch_calibration.txt confirms for 16-byte gathers — + WRITE_SIZE, KiB), the VALU line and what
    the waves spend their cycles on, all per round of the child run and scaled by `rays_scale` (this run's rays per launch / the child's:
    the per-ray work of a deterministic workload does not depend on the wavefront size). `ms_launch` is THIS run's HIP-event launch time.
    The class the data put the stage in — `limiter_class`: "hbm" when the measured traffic is at least half of the 8 TB/s peak in the
    launch's time, else "valu" when the VALU instructions fill at least 0.6 of the issue cycles, else "latency" (neither the memory
    system nor the issue slots are at their limit: the waves wait)."""
    # per round = the stage's sums / the rounds the child ran (its warm-up, timed steps and counter replay are the same iterations, so the
    # mean over all rounds is the timed rounds' mean). A traversal stage launches ONE of its kernels per round (the plain instantiation, or
    # the SORTED one from round 1 on under IGD_RAY_SORT): rounds = the launches of all of them; the shading stage launches each of its
    # kernels once per round: rounds = the launches of the most launched one. `tot` (sums) serves the ratios.
    tot, launches = {}, {}
    for k, cs in ctr.items():
        if any(k.startswith(p) for p in STAGE_KERNELS[stage]):
            for c, v in cs.items():
                tot[c] = tot.get(c, 0.0) + v["sum"]
                launches.setdefault(c, []).append(v["launches"])
    if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
        return None
    per_round = {c: tot[c] / max(1, sum(n) if stage.startswith("k_traverse") else max(n)) for c, n in launches.items()}
    traffic = (2.0 * per_round["FETCH_SIZE"] + per_round["WRITE_SIZE"]) * 1024.0 * rays_scale
    secs = ms_launch * 1e-3
    ev = {"traffic": int(traffic), "measured_frac": round(traffic / secs / 1e9 / HBM_PEAK_GBS, 5) if secs > 0 else None}
    if all(tot.get(c, 0) > 0 for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")):
        cpi = (valu_cycles_per_inst() or 3.0) if stage.startswith("k_traverse") else SHADE_CYCLES_PER_INST
        insts = per_round["SQ_INSTS_VALU"] * rays_scale
        util = tot["SQ_THREAD_CYCLES_VALU"] / (64.0 * tot["SQ_ACTIVE_INST_VALU"])
        g = insts * 64.0 * util / secs / 1e9 if secs > 0 else 0.0
        ev["valu"] = {"achieved": round(g, 1), "peak": round(VALU_PEAK_GLANE_OPS, 1), "unit": "G lane-ops/s", "frac": round(g / VALU_PEAK_GLANE_OPS, 4),
                      "insts_per_launch": int(insts),
                      "issue_frac": round(insts * cpi / (1024 * SHADER_GHZ * 1e9 * secs), 4) if secs > 0 else None,
                      "issue_frac_note": f"{cpi} cycles per wave64 instruction ("
                                         + ("the closest-hit traversal's dynamic opcode histogram x calibrated prices, profiles/*_issue_accounting.json" if stage.startswith("k_traverse")
                                            else "the lean shading kernel's mix, DESIGN.md 4.3") + f"), 1024 SIMDs at {SHADER_GHZ} GHz under load"}
        ev["limiter"] = {"valu_lane_utilisation": round(util, 4), "wave_wait_share": round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 4),
                         "wave_issue_share": round(tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 4) if tot.get("SQ_ACTIVE_INST_ANY") else None,
                         "source": "SQ counters of this run's rocprofv3 --pmc pass"}
    mf, isf = ev["measured_frac"] or 0.0, (ev.get("valu") or {}).get("issue_frac") or 0.0
    ev["limiter_class"] = "hbm" if mf >= 0.5 else ("valu" if isf >= 0.6 else "latency")
    return ev


def live_traffic(args, rays_per_launch):
    """roofline.traffic and the VALU / limiter blocks of the headline kernel measured in THIS run: three passes of the same command under the
    counters (pmc_passes: FETCH_SIZE, WRITE_SIZE, the SQ group), priced per ray x this run's rays per launch."""
    # (the per-ray figures of a deterministic workload do not depend on the wavefront size: a long run is profiled on a shorter one; the
    # child's warm-up is the same batch of iterations as its timed steps, so the mean over ALL its launches is the timed launches' mean)
    pmc_steps = min(args.steps, 32)
    ctr, line = pmc_passes(args.scene, args.width, args.height, args.spi, pmc_steps, [["FETCH_SIZE"], ["WRITE_SIZE"], SQ_GROUP])
    if ctr is None:
        return None, line
    n_rays = line["rays"]["camera"] + line["rays"]["bounce"]
    child_rpl = n_rays / max(1, line["roofline"]["launches"])
    return (ctr, line, rays_per_launch / child_rpl), ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* --kernel-trace, own passes of this command as child "
                                                      f"processes at {pmc_steps} steps; k_traverse<closest> sums per round, reads x 2 (gfx950 correction, "
                                                      "profiles/r05_fetch_calibration.txt), per-ray figure x this run's rays per launch")


def shade_stream_bytes(n_in, n_bounce, n_shadow):
    """SURVEY.md 8(d) / DESIGN.md 3, the reference's columns: a hit read (rayA rayB meta pay hit + eta hit_v = 88 B), a continuation ray
    written (rayA rayB meta pay + eta = 68 B), a shadow ray written (48 B). Accumulator updates and the scene's shading tables come on top
    and are not priced: a lower bound of the SURVEY model. What the kernels move since round 5 is less: moved_bytes()."""
    return 88 * n_in + 68 * n_bounce + 48 * n_shadow


def shadow_stream_bytes(n_shadow, n_unoccluded):
    """56 B per shadow ray (the 48 B the any-hit launch reads + its id) and 24 B per unoccluded splat."""
    return 56 * n_shadow + 24 * n_unoccluded


def moved_bytes(stats, sb):
    """Stream bytes the three stages move BY CONSTRUCTION since round 5 (the SURVEY model above prices columns the kernels no
    longer carry): the columns each stream kind holds (`sb` = igd_stats.stream_bytes, include/igd_device.h) x the ray counts. k_shade:
    every hit's columns (a miss that needs no shading reads its hit only: an upper bound there), bounce and shadow rays written;
    accumulator updates of k_shade are not priced."""
    cam, bounce, shadow, unocc = stats["camera_rays"], stats["bounce_rays"], stats["shadow_rays"], stats["unoccluded"]
    return {"k_traverse<closest>": cam * (sb[0] + sb[2]) + bounce * (sb[1] + sb[2]),
            "k_shade": cam * sb[3] + bounce * sb[4] + bounce * sb[5] + shadow * sb[6],
            "k_traverse<any>": shadow * sb[6] + unocc * sb[7]}


def measure_config(name, scene_path, W, H, spi, steps, warmup, capacity, device_index=0, live=True):
    """One more workload of BASELINE.json's `configs` through the same product path as the headline (igd_render per iteration on one
    GPU, inputs resident, HIP-event stage timers on the render stream), with the roofline of ITS dominant kernel: the stage with the
    most GPU time among closest-hit traversal, shading and any-hit traversal, algorithmic stream bytes per launch from a counter
    replay of the same steps / that stage's average launch time — and, measured in the run (pmc_passes: three child passes of this
    workload under rocprofv3 --pmc), that stage's HBM traffic, VALU line and wave states, from which `bound` is decided."""
    from ignis_amd import Device, LoadedScene
    t_load = time.perf_counter()
    scene = LoadedScene.from_file(scene_path, W, H)
    t_load = time.perf_counter() - t_load
    dev = Device(device_index, acquire_stats=1, stream_capacity=capacity)
    dev.assign_scene(scene)
    dev.resize(W, H)

    def run(on, k):
        for it in range(k):
            on.render(spi, W, H, iteration=it, seed=SEED)
    run(dev, warmup)
    dev.synchronize()
    dev.clear_framebuffer()
    dev.reset_stats()
    t0 = time.perf_counter()
    run(dev, steps)
    dev.synchronize()
    elapsed = time.perf_counter() - t0
    st = dev.stats()
    dev.close()
    cdev = Device(device_index, acquire_stats=2, stream_capacity=capacity)
    cdev.assign_scene(scene)
    cdev.resize(W, H)
    run(cdev, steps)
    cs = cdev.stats()
    node_bytes = cdev.node_bytes()
    cdev.close()
    rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    n_primary = cs["camera_rays"] + cs["bounce_rays"]
    rounds = max(1, st["traverse_primary_launches"])
    rounds2 = max(1, st["traverse_secondary_launches"])
    geom_resident = int(scene.scene.primbvh_size) + int(scene.scene.scene_node_count) * 256 + int(scene.scene.scene_leaf_count) * 96
    per_launch = {
        "k_traverse<closest>": (st["ms_traverse_primary"] / rounds,
                                stream_bytes(n_primary) / rounds + min(geometry_bytes(cs["nodes_primary"], cs["tris_primary"], cs["leaves_primary"], node_bytes) / rounds, geom_resident)),
        "k_shade": (st["ms_shade"] / rounds, shade_stream_bytes(n_primary, cs["bounce_rays"], cs["shadow_rays"]) / rounds),
        "k_traverse<any>": (st["ms_traverse_secondary"] / rounds2,
                            shadow_stream_bytes(cs["shadow_rays"], cs["unoccluded"]) / rounds2
                            + min(geometry_bytes(cs["nodes_secondary"], cs["tris_secondary"], cs["leaves_secondary"], node_bytes) / rounds2, geom_resident)),
    }
    kernel = max(per_launch, key=lambda k: per_launch[k][0])
    ms, alg = per_launch[kernel]
    achieved = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    moved = moved_bytes(cs, cs["stream_bytes"])[kernel] / (rounds2 if kernel == "k_traverse<any>" else rounds)
    roof = {"bound": None, "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None, "measured_frac": None, "avg_launch_ms": round(ms, 5), "launches": int(rounds), "algorithmic_bytes_per_launch": int(alg),
            "moved_stream_bytes_per_launch": int(moved),
            "note": "dominant stage of this workload by HIP-event time; `achieved` = the SURVEY 8(d) byte model (the reference's columns: 60 B per traversed ray, 88 / 68 / 48 B "
                    "per shaded hit / bounce ray / shadow ray, + min(geometry visited, resident) for the traversals) per launch / average launch time; "
                    "`moved_stream_bytes_per_launch` = the columns the kernels carry since round 5 x the ray counts (igd_stats.stream_bytes); `traffic` = HBM-side bytes "
                    "per launch measured in this run (2 x FETCH_SIZE + WRITE_SIZE of the stage's kernels); `bound` = the class the counters put the stage in "
                    "(stage_evidence: hbm / valu / latency); k_shade = the sort passes + every class kernel of a round"}
    why = None
    if live:
        ctr, line = pmc_passes(scene_path, W, H, spi, steps, [["FETCH_SIZE"], ["WRITE_SIZE"], SQ_GROUP])
        if ctr is None:
            why = line
        else:
            ev = stage_evidence(ctr, line, kernel, ms)
            if ev is None:
                why = f"no {kernel} launches under the counters"
            else:
                roof.update({"traffic": ev["traffic"], "measured_frac": ev["measured_frac"], "valu": ev.get("valu"), "limiter": ev.get("limiter"),
                             "bound": ev["limiter_class"], "limiter_class": ev["limiter_class"],
                             "traffic_source": f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* --kernel-trace, own passes of this workload at {steps} steps"})
    if roof["bound"] is None:
        roof["bound"] = "unknown"
        roof["traffic_source"] = f"none: {why or 'live measurement switched off'}"
    return {"name": name,
            "workload": f"{name}: {W}x{H}, path integrator, spi {spi} x {steps} iterations, seed {SEED}",
            "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "msamples_per_s": round(st["camera_rays"] / elapsed / 1e6, 3), "timed_seconds": round(elapsed, 3), "scene_load_seconds": round(t_load, 2),
            "rays": {"camera": st["camera_rays"], "bounce": st["bounce_rays"], "shadow": st["shadow_rays"]},
            "stage_ms": {k: round(st[k], 3) for k in ("ms_generate", "ms_traverse_primary", "ms_shade", "ms_traverse_secondary", "ms_tail", "ms_resolve", "ms_ray_sort")},
            "geometry_resident_bytes": geom_resident,
            "roofline": roof}


def extra_configs(args):
    """BASELINE.json configs 3 and 4 next to the headline: the seeded 1 M-triangle stand-in with the divergent
    material mix (the Bedroom asset is absent; tools/make_standin_scene.py, generated here into a temporary directory) and
    scenes/many_point_lights.json, both 1920x1080."""
    import subprocess
    import tempfile
    out = []
    steps, warmup = 16, 16
    # (streams as for the headline: a whole batch of iterations is one wavefront; the headline's device is closed by now)
    cap = int(os.environ.get("BENCH_CAPACITY", 1 << 29))
    with tempfile.TemporaryDirectory(prefix="standin_") as tmp:
        t = time.perf_counter()
        made = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_standin_scene.py"), tmp, "--triangles", "1000000", "--instances", "96", "--seed", "7",
                               "--materials", "divergent"], capture_output=True, text=True)
        if made.returncode == 0:
            c = measure_config("config 3 stand-in (seeded procedural scene, 1 M unique triangles, 32 divergent materials, 4 area lights; NOT the Bedroom asset)",
                               os.path.join(tmp, "standin.json"), WIDTH, HEIGHT, SPI, steps, warmup, cap, live=not args.no_live_traffic)
            c["generator"] = "tools/make_standin_scene.py --triangles 1000000 --instances 96 --seed 7 --materials divergent"
            c["generate_seconds"] = round(time.perf_counter() - t - c["scene_load_seconds"] - c["timed_seconds"], 2)
            out.append(c)
        else:
            out.append({"name": "config 3 stand-in", "error": made.stderr[-400:]})
    out.append(measure_config("config 4 scenes/many_point_lights.json", os.path.join(ROOT, "scenes", "many_point_lights.json"), WIDTH, HEIGHT, SPI, 32, 32, cap, live=not args.no_live_traffic))
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")

    dist = None
    torch = None
    distributed = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))  # BENCH_FORCE_DIST=1: exercise the RCCL path with a single rank
    native = distributed and args.dist == "rccl"
    if distributed and not native:
        # torch first: its bundled HIP runtime and RCCL are the ones every library in this process binds to
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if native:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")

    import numpy as np
    from ignis_amd import Device, LoadedScene, sharding

    W, H, spi = args.width, args.height, args.spi
    scene = LoadedScene.from_file(args.scene, W, H)
    # streams sized once for a full batch of iterations (2^29 camera rays, ~157 GB of the 288 GB): the warm-up then pays for the
    # allocation (and for the driver's scrubbing of memory a previous process just released), not the timed region
    CAPACITY = int(os.environ.get("BENCH_CAPACITY", 1 << 29))
    dev = Device(local_rank, acquire_stats=0 if args.no_stage_timers else 1, stream_capacity=CAPACITY)
    dev.assign_scene(scene)
    dev.resize(W, H)

    comm = None
    if native:
        # The device library's own communicator: ncclCommInitRank inside libig_device_hip.so, the id handed round by ignis_amd.comm.
        # Brought up by a vote (ignis_amd/comm.py agree): every rank says whether it can take part before anyone enters
        # ncclCommInitRank, every rank reports whether its communicator came up and talks, rank 0 broadcasts the verdict — so the ranks
        # either ALL go on natively or ALL start over on the torch.distributed process group, and no rank is left inside a collective
        # the others abandoned (a fresh process each: torch has to be the first to load its HIP runtime and RCCL).
        from ignis_amd.comm import Comm
        fail = os.environ.get("BENCH_NATIVE_COMM_FAIL")  # tests: "raise" / "hang" / "probe" (or any other value = "raise"), on the rank BENCH_NATIVE_COMM_FAIL_RANK names (default: every rank)
        if fail and int(os.environ.get("BENCH_NATIVE_COMM_FAIL_RANK", rank)) != rank:
            fail = None
        if fail and fail not in ("raise", "hang", "probe"):
            fail = "raise"
        comm = Comm.agreed(dev, rank, world, deadline=float(os.environ.get("BENCH_COMM_DEADLINE", "60")), fail=fail)
        if comm is None and Comm.last_fallback_reason.startswith("abort: "):
            raise SystemExit(f"[bench] rank {rank}: {Comm.last_fallback_reason} — the job is incomplete, no collective can run")
        if comm is None:
            print(f"[bench] rank {rank}: the ranks agreed not to use the native RCCL communicator ({Comm.last_fallback_reason}); re-executing with --dist torch", file=sys.stderr, flush=True)
            argv = [a for i, a in enumerate(sys.argv) if not (a == "--dist" or (i > 0 and sys.argv[i - 1] == "--dist") or a.startswith("--dist="))]
            os.environ.pop("BENCH_NATIVE_COMM_FAIL", None)
            sys.stderr.flush()
            # (no dev.close(): a helper thread may still sit inside ncclCommInitRank on this device; the new process image disposes of both)
            os.execv(sys.executable, [sys.executable] + argv + ["--dist", "torch"])

    def barrier():
        if comm is not None:
            comm.barrier()  # (an all-reduce on the render stream, waited for)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    shards = args.as_rank_of if args.as_rank_of > 1 else world
    by_rows = args.sharding == "rows"  # N = 1: rows of a single shard = the whole film, same code path
    # One igd_render per iteration, like Runtime::step. The device executes consecutive iterations as one wavefront
    # (up to 2^29 camera rays, bit-identical to executing them one by one): that is what keeps a
    # row-sharded rank, which owns 1 / N of every iteration, as efficient as a whole film on one GPU.
    steps_per_rank = max(args.steps, args.warmup)  # iterations sharding: a rank's iterations are consecutive

    def step(on, it):
        if by_rows:
            on.render(spi, W, H, iteration=it, seed=SEED, row_offset=rank if shards > 1 else 0, row_stride=shards)
        else:
            on.render(spi, W, H, iteration=rank * steps_per_rank + it, seed=SEED)  # ignis_amd.sharding.shard_iterations

    def run(on, steps):
        for it in range(steps):
            t = time.perf_counter()
            step(on, it)
            if os.environ.get("BENCH_TRACE") and time.perf_counter() - t > 0.005:
                print(f"[trace] call {it}: {(time.perf_counter() - t) * 1e3:.1f} ms", file=sys.stderr, flush=True)

    run(dev, args.warmup)
    dev.synchronize()
    dev.clear_framebuffer()
    dev.reset_stats()

    fb_tensor = None
    collective = None
    if comm is not None:
        if not by_rows:
            raise SystemExit("--sharding iterations needs --dist torch (the native communicator implements the gather of owned rows)")
        # RCCL sets up its channels / kernels for a message size on first use: do that outside the timed region (the film is cleared below)
        comm.gather_rows(dst=0)
        dev.clear_framebuffer()
        collective = {"op": "gather of owned rows to rank 0 (grouped ncclSend / ncclRecv by libig_device_hip.so, igd_comm_gather_rows)",
                      "bytes_per_rank": sharding.gather_bytes(H, W, world), "backend": "rccl (dlopen by the device library, no torch)",
                      "world_size_from_backend": comm.world_size_from_backend()}
    if dist is not None:
        class _Wrap:  # zero-copy view of the device framebuffer for RCCL
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}
        fb_tensor = torch.as_tensor(_Wrap(dev.framebuffer_device_ptr(), (H, W, 3)), device=torch.device("cuda", local_rank))
        use_gather = by_rows and args.collective == "gather"

        def collective_op(t):
            if use_gather:
                sharding.gather_rows(t, rank, world, dist, dst=0)
            else:
                dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        # RCCL sets up its channels / kernels for a message size on first use: do that outside the timed region
        scratch = torch.zeros_like(fb_tensor)
        collective_op(scratch)
        torch.cuda.synchronize()
        del scratch
        collective = {"op": "gather of owned rows to rank 0" if use_gather else "reduce(SUM) of whole framebuffers to rank 0",
                      "bytes_per_rank": sharding.gather_bytes(H, W, world) if use_gather else W * H * 12,
                      "backend": dist.get_backend(), "world_size_from_backend": dist.get_world_size()}
    barrier()
    t0 = time.perf_counter()
    run(dev, args.steps)  # calls return at once or when a wavefront's rounds are done; tails + resolves overlap the next one
    t_sync = time.perf_counter()
    dev.synchronize()  # everything submitted above is finished before the clock stops (and before the collective)
    if os.environ.get("BENCH_TRACE"):
        print(f"[trace] loop {(t_sync - t0) * 1e3:.1f} ms, final synchronize {(time.perf_counter() - t_sync) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    if comm is not None:
        comm.gather_rows(dst=0)  # the ONLY collective: final accumulation of the tile-sharded framebuffer on rank 0 (returns when it is complete)
    if dist is not None:
        # the ONLY collective: final accumulation of the tile-sharded framebuffer on rank 0
        collective_op(fb_tensor)
        torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    st = dev.stats()
    rays_local = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    samples_local = st["camera_rays"]
    if comm is not None:
        elapsed = comm.allreduce([elapsed], "max")[0]
        rays_total, samples_total = comm.allreduce([rays_local, samples_local], "sum")
    elif dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rays_local, samples_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays_total, samples_total = float(c[0].item()), float(c[1].item())
    else:
        rays_total, samples_total = float(rays_local), float(samples_local)

    # ---- BASELINE.json's literal configuration: 64 spp = 8 steps, started on an idle device (no batching across more than
    # those 8 iterations, nothing of a previous wavefront to overlap with) — reported next to the steady-state figure
    literal = None
    if world == 1 and not args.no_literal_config and shards == 1:
        dev.clear_framebuffer()
        dev.reset_stats()
        l0 = time.perf_counter()
        run(dev, 8)
        dev.synchronize()
        l_el = time.perf_counter() - l0
        ls = dev.stats()
        literal = {"workload": f"{W}x{H}, spi {spi} x 8 iterations = {8 * spi} spp, idle device to finished image",
                   "value": round((ls["camera_rays"] + ls["bounce_rays"] + ls["shadow_rays"]) / l_el / 1e6, 3), "unit": "Mrays/s",
                   "msamples_per_s": round(ls["camera_rays"] / l_el / 1e6, 3), "seconds": round(l_el, 4)}

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (closest-hit traversal), rank 0's launches
        launches = max(1, st["traverse_primary_launches"])
        avg_ms = st["ms_traverse_primary"] / launches
        # units per launch: replay one batch of the same steps with the work counters on (deterministic workload; the per-launch
        # averages of a run that is a whole number of 32-iteration batches equal those of one batch)
        replay = 32 if (args.steps % 32 == 0 and args.steps >= 32 and (W, H, spi) == (WIDTH, HEIGHT, SPI) and shards == 1) else args.steps
        dev.close()  # (its streams would not fit beside a second set at the larger capacities)
        cdev = Device(local_rank, acquire_stats=2, stream_capacity=CAPACITY)
        cdev.assign_scene(scene)
        cdev.resize(W, H)
        run(cdev, replay)
        cs = cdev.stats()
        node_bytes = cdev.node_bytes()
        cdev.close()
        n_primary = cs["camera_rays"] + cs["bounce_rays"]
        c_launches = max(1, cs["traverse_primary_launches"])
        s_per_launch = stream_bytes(n_primary) / c_launches
        g_per_launch = geometry_bytes(cs["nodes_primary"], cs["tris_primary"], cs["leaves_primary"], node_bytes) / c_launches
        # Each byte of the geometry has to come from HBM at least once per launch; what the rays re-visit beyond that is
        # served by L1 / L2 / Infinity Cache whenever the BVH fits them. The HBM-side algorithmic bytes are therefore
        # the streams + min(geometry visited, geometry resident); the 8(d) figure incl. every re-visit is kept separately.
        geom_resident = int(scene.scene.primbvh_size) + int(scene.scene.scene_node_count) * 256 + int(scene.scene.scene_leaf_count) * 96
        if node_bytes == 128:  # the kernels read the 128-byte records, not the Node8 tables: half of the node bytes are resident for them
            geom_resident -= (_primbvh_nodes(scene) + int(scene.scene.scene_node_count)) * 128
        hbm_alg = s_per_launch + min(g_per_launch, geom_resident)
        achieved = hbm_alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        incl_cache = (s_per_launch + g_per_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # measured HBM-side bytes per launch of the same kernel: PMC passes of this command, summarised into
        # profiles/ by tools/prof_summary.py (counters cannot be read from inside the process)
        traffic, traffic_src, limiter, valu = None, None, None, None
        # the committed counter summary of this workload (the fallback of the live measurement): the headline's own file, or the one
        # --profile-key names (keyed by the caller, not by the scene's basename: two stand-ins are both called standin.json)
        key = args.profile_key if args.profile_key else (None if args.scene != SCENE else "")
        tname = None if key is None else (f"{PROFILE_TAG}_traffic.json" if key == "" else f"{PROFILE_TAG}_traffic_{key}.json")
        if tname and not os.path.exists(os.path.join(ROOT, "profiles", tname)):
            prev = tname.replace(PROFILE_TAG + "_", "r05_", 1)
            tname = prev if os.path.exists(os.path.join(ROOT, "profiles", prev)) else tname
        tpath = os.path.join(ROOT, "profiles", tname) if tname else None
        rays_per_launch = n_primary / c_launches
        if tpath and os.path.exists(tpath) and (W, H, spi) == (WIDTH, HEIGHT, SPI) and world == 1:
            tj = json.load(open(tpath))
            tk = next((v for k, v in tj["kernels"].items() if k.startswith("k_traverse<false, false, false")), None)  # <closest, no stats, not DEEP[, no spheres]>
            pr = tj.get("closest_hit_per_ray")
            if tk and (pr or tj.get("steps") == args.steps):
                # The counters were collected on this command at tj["steps"] steps. Per launch they scale with the rays a launch
                # traverses (the per-ray work of a deterministic workload does not depend on the wavefront size), so a run at another
                # step count prices its own launches with the profiled per-ray figures.
                scale = (rays_per_launch / pr["rays_per_launch_profiled"]) if pr else 1.0
                traffic = int(pr["hbm_bytes"] * rays_per_launch) if pr else int(tk["hbm_bytes"])
                traffic_src = (f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, own passes of this command at {tj.get('steps')} steps, x2 read correction; "
                               f"per-ray figure x the {rays_per_launch:.0f} rays per launch of this run)")
                if "valu_lane_utilisation" in tk:
                    limiter = {"valu_lane_utilisation": tk["valu_lane_utilisation"], "wave_wait_share": tk.get("wave_wait_share"),
                               "wave_issue_share": tk.get("wave_issue_share"), "source": f"profiles/{tname} (SQ counters)"}
                if tk.get("valu_lane_ops_per_launch"):
                    # VALU line: wave64 VALU instructions per second against the issue slots of 1024 SIMDs; a slot is priced at the 2.4 - 4.4
                    # cycles profiles/r04_valu_calibration.txt measured, weighted with the kernel's dynamic opcode histogram (valu_cycles_per_inst())
                    insts = tk["valu_insts_per_launch"] * scale
                    g = tk["valu_lane_ops_per_launch"] * scale / (avg_ms * 1e-3) / 1e9
                    cpi = valu_cycles_per_inst()
                    valu = {"achieved": round(g, 1), "peak": round(VALU_PEAK_GLANE_OPS, 1), "unit": "G lane-ops/s", "frac": round(g / VALU_PEAK_GLANE_OPS, 4),
                            "insts_per_ray": round(insts * 64.0 / rays_per_launch, 1),
                            "issue_frac": round(insts * cpi / (1024 * SHADER_GHZ * 1e9 * avg_ms * 1e-3), 4) if cpi else None,
                            "issue_frac_note": f"{cpi} cycles per wave64 instruction (profiles/*_issue_accounting.json: dynamic opcode histogram x calibrated prices), {SHADER_GHZ} GHz under load",
                            "source": f"profiles/{tname}"}
        traffic_file, traffic_file_src = traffic, traffic_src
        limiter_class = None
        if world == 1 and shards == 1 and not distributed and not args.no_live_traffic and not args.no_stage_timers:
            live, why = live_traffic(args, rays_per_launch)
            ev = stage_evidence(live[0], live[1], "k_traverse<closest>", avg_ms, live[2]) if live else None
            if ev:
                traffic, traffic_src = ev["traffic"], why
                if ev.get("valu"):
                    valu = dict(ev["valu"], insts_per_ray=round(ev["valu"]["insts_per_launch"] * 64.0 / rays_per_launch, 1), source="this run's SQ pass")
                    limiter = ev["limiter"]
                limiter_class = ev["limiter_class"]
            elif traffic_src:
                traffic_src += f" [live measurement skipped: {why if not live else 'no closest-hit launches under the counters'}]"
            else:
                traffic_src = f"none: {why}; no committed counter summary for this workload" + ("" if tname else " (--profile-key names one)")
        if limiter_class is None and traffic is not None and avg_ms > 0:
            mf, isf = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, (valu or {}).get("issue_frac") or 0.0
            limiter_class = "hbm" if mf >= 0.5 else ("valu" if isf >= 0.6 else "latency")
        moved = moved_bytes(cs, cs["stream_bytes"])["k_traverse<closest>"] / c_launches
        roofline = {"bound": "hbm", "kernel": "k_traverse<closest>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "measured_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and avg_ms > 0 else None,
                    "note": "achieved = HBM-side algorithmic bytes per launch by SURVEY 8(d)'s model (60 B per ray of the reference's stream columns + the geometry bytes that "
                            "have to be fetched at least once: min(visited, resident)) / launch time — the MODEL, not the bytes moved: `moved_stream_bytes_per_launch` prices the "
                            "columns the kernels carry since round 5 (igd_stats.stream_bytes x ray counts); `incl_cache_hits` prices every Node8 / triangle / leaf visit at its size "
                            "(SURVEY 8d's A), most of which L1 / L2 serve when the BVH is small; `traffic` = measured HBM-side bytes per launch; `bound` names the axis of "
                            "achieved / peak, `limiter_class` what the counters say the kernel waits for (stage_evidence)",
                    "incl_cache_hits": {"achieved": round(incl_cache, 2), "unit": "GB/s", "bytes_per_launch": int(s_per_launch + g_per_launch)},
                    "stream_bytes_per_launch": int(s_per_launch), "moved_stream_bytes_per_launch": int(moved), "stream_bytes_per_ray": cs["stream_bytes"],
                    "geometry_resident_bytes": geom_resident,
                    "node_bytes": node_bytes, "traffic_from_profiles": traffic_file,
                    "valu": valu, "limiter": limiter, "limiter_class": limiter_class, "avg_launch_ms": round(avg_ms, 5), "launches": int(launches),
                    "algorithmic_bytes_per_launch": int(hbm_alg)}

        stage_ms = {k: round(st[k], 3) for k in ("ms_generate", "ms_traverse_primary", "ms_shade", "ms_traverse_secondary", "ms_tail", "ms_resolve", "ms_ray_sort")}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            # bounded sample of the same workload: whole iterations until ~12 s of wall clock (at most 8)
            cw, ch = W, H
            cpu_rays = cpu_samples = 0
            n_it, threads = 0, 1
            use, hw, quota = effective_cpus()
            t1 = time.perf_counter()
            while n_it < 8 and (n_it == 0 or time.perf_counter() - t1 < 12.0):
                _, os_ = oracle.render(scene, spi, cw, ch, iteration=n_it, seed=SEED, threads=use)
                cpu_rays += os_["camera_rays"] + os_["bounce_rays"] + os_["shadow_rays"]
                cpu_samples += os_["camera_rays"]
                threads = int(os_["threads_used"])
                n_it += 1
            dt = time.perf_counter() - t1
            # the same restatement on ONE thread over a quarter-size film (the all-thread figure divided by it = how the port scales
            # on this host; before round 6 its per-thread counters shared cache lines and 8 threads ran 1.9 x one)
            t2 = time.perf_counter()
            _, o1 = oracle.render(scene, spi, max(16, cw // 4), max(16, ch // 4), iteration=0, seed=SEED, threads=1)
            dt1 = time.perf_counter() - t2
            one = (o1["camera_rays"] + o1["bounce_rays"] + o1["shadow_rays"]) / dt1 / 1e6
            cpu = {"value": round(cpu_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "host": f"{hw} hardware threads visible" + (f", container CPU quota {quota:g}" if quota else ", no CPU quota") + f": {use} worker threads",
                   "single_thread": {"value": round(one, 3), "unit": "Mrays/s", "sample": f"1 iteration at {max(16, cw // 4)}x{max(16, ch // 4)} spi {spi}", "seconds": round(dt1, 2)},
                   "speedup_over_single_thread": round(cpu_rays / dt / 1e6 / one, 2) if one > 0 else None,
                   "sample": f"{n_it} iteration(s) of {os.path.basename(args.scene)} {cw}x{ch} spi {spi} (oracle/, CPU restatement of cpu_trace, not the AnyDSL binary)",
                   "msamples_per_s": round(cpu_samples / dt / 1e6, 3), "seconds": round(dt, 2)}

        total_iterations = args.steps * (1 if by_rows else world)
        out = {
            "metric": "Mrays/s (primary+shadow)",
            "value": round(rays_total / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if by_rows else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{os.path.relpath(args.scene, ROOT)} {W}x{H}, path integrator, spi {spi} x {total_iterations} iterations, seed {SEED}",
                       "sharding": "whole film" if shards == 1 else (f"film rows interleaved over {shards} GPUs (tile-sharded film), one RCCL collective at the end" if by_rows else
                                                                    f"{args.steps} full-film iterations per GPU (rank r: iterations r*K .. r*K+K-1) + one RCCL reduce")},
            "timed_seconds": round(elapsed, 3),
            "msamples_per_s": round(samples_total / elapsed / 1e6, 3),
            "literal_config": literal,
            "collective": collective,
            "rays": {"camera": st["camera_rays"], "bounce": st["bounce_rays"], "shadow": st["shadow_rays"], "scope": "rank 0"},
            "stage_ms_rank0": stage_ms,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if world == 1 and shards == 1 and not args.no_extra_configs and args.scene == SCENE:
            out["configs"] = extra_configs(args)

    if comm is not None:
        pass  # (the communicator went with its device: rank 0 closed that before the counter replay)
    dev.close()  # (idempotent: rank 0 closed it before the counter replay)
    # RCCL writes a version banner to the C stdout buffer of the ranks; every rank pushes its buffer out before rank 0
    # prints, so that the JSON line is the LAST line of the job's stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
