#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric for the Ignis hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one render iteration (Runtime::step, src/runtime/Runtime.cpp:334-387) of the workload
BASELINE.json's metric is quoted on: scenes/diamond_scene.json, 1920x1080, path integrator,
spi 8 (64 spp = 8 steps). Inputs (scene tables) are resident in HBM before the timed region.
With N > 1 the camera samples are sharded with no data-path exchange and the framebuffers are
reduced to rank 0 over RCCL once, inside the timed region. Default partition: whole-film iterations (rank r renders
iterations r*K .. r*K+K-1: K steps per GPU, N x K iterations in total, "weak"). `--sharding rows` tile-shards the film
instead (rank r renders rows r, r+N, ... of every iteration; the device batches the small per-rank iterations into
full-size wavefronts; the sum of the shards is the single-GPU image bit for bit; "strong": K iterations in total,
which leaves each of 8 GPUs only K / 8 iterations' worth of work — 6.4x at K = 16, 7.9x from K = 64).

Prints ONE JSON line (rank 0): Mrays/s = (camera + bounce + shadow rays) / s as the reference counts
them (src/runtime/Statistics.cpp:286-290), plus Msamples/s (src/frontend/cli/main.cpp:134), the
roofline of the dominant kernel (closest-hit traversal) and the CPU baseline (oracle) on rank 0.
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
TRAFFIC_FILE = "r01_traffic.json"  # PMC summary of this command, see tools/collect_profiles.sh


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timers", action="store_true", help="experiments only: no HIP-event stage timers in the timed loop (roofline.achieved becomes 0)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spi", type=int, default=SPI)
    ap.add_argument("--sharding", choices=("iterations", "rows"), default="iterations", help="N > 1: how camera samples are split")
    ap.add_argument("--as-rank-of", type=int, default=0, help="experiments only: one process renders what rank 0 of N row-sharding ranks would (estimate of per-GPU throughput at N GPUs)")
    ap.add_argument("--scene", default=SCENE, help="other scene file (not the headline workload), e.g. tools/make_standin_scene.py output")
    return ap.parse_args()


def algorithmic_bytes(n_rays, nodes, tris, leaves):
    """SURVEY.md 8(d) with this backend's layouts: 60 B per closest-hit ray (40 B read + 20 B hit
    written) + 256 B per Node8 fetched + 52 B per triangle tested (a 208 B Tri4 packet holds 4)
    + 96 B per EntityLeaf1 tested."""
    return 60 * n_rays + 256 * nodes + 52 * tris + 96 * leaves


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
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST=1: exercise the RCCL path with a single rank
        # torch first: its bundled HIP runtime and RCCL are the ones every library in this process binds to
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import numpy as np
    from ignis_amd import Device, LoadedScene

    W, H, spi = args.width, args.height, args.spi
    scene = LoadedScene.from_file(args.scene, W, H)
    # streams sized once for a full batch of iterations (2^28 camera rays, ~78 GB): the warm-up then pays for the
    # allocation (and for the driver's scrubbing of memory a previous process just released), not the timed region
    CAPACITY = int(os.environ.get("BENCH_CAPACITY", 1 << 28))
    dev = Device(local_rank, acquire_stats=0 if args.no_stage_timers else 1, stream_capacity=CAPACITY)
    dev.assign_scene(scene)
    dev.resize(W, H)

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    by_rows = (world > 1 and args.sharding == "rows") or args.as_rank_of > 1
    shards = args.as_rank_of if args.as_rank_of > 1 else world
    # One igd_render per iteration, like Runtime::step. The device executes consecutive iterations as one wavefront
    # (up to 2^28 camera rays, bit-identical to executing them one by one): that is what keeps a
    # row-sharded rank, which owns 1 / N of every iteration, as efficient as a whole film on one GPU.

    steps_per_rank = max(args.steps, args.warmup)  # a rank's iterations are consecutive, so the device can batch them

    def step(on, it):
        if by_rows:
            on.render(spi, W, H, iteration=it, seed=SEED, row_offset=rank, row_stride=shards)
        else:
            on.render(spi, W, H, iteration=rank * steps_per_rank + it, seed=SEED)  # ignis_amd.sharding.shard_iterations

    def run(on, steps):
        for it in range(steps):
            t = time.perf_counter()
            step(on, it)
            if os.environ.get("BENCH_TRACE") and time.perf_counter() - t > 0.005:
                print(f"[trace] call {it}: {(time.perf_counter() - t) * 1e3:.1f} ms", file=sys.stderr, flush=True)

    run(dev, args.warmup)
    warm = dev.stats()  # (profilers see the warm-up launches too: their average is reported next to the timed one)
    dev.clear_framebuffer()
    dev.reset_stats()

    fb_tensor = None
    if dist is not None:
        class _Wrap:  # zero-copy view of the device framebuffer for RCCL
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}
        fb_tensor = torch.as_tensor(_Wrap(dev.framebuffer_device_ptr(), (H, W, 3)), device=torch.device("cuda", local_rank))

    if dist is not None:
        # RCCL sets up its channels / kernels for a message size on first use: do that outside the timed region
        scratch = torch.zeros_like(fb_tensor)
        dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        del scratch
    barrier()
    t0 = time.perf_counter()
    run(dev, args.steps)  # calls return at once or when a wavefront's rounds are done; tails + resolves overlap the next one
    t_sync = time.perf_counter()
    dev.synchronize()  # everything submitted above is finished before the clock stops (and before the reduce)
    if os.environ.get("BENCH_TRACE"):
        print(f"[trace] loop {(t_sync - t0) * 1e3:.1f} ms, final synchronize {(time.perf_counter() - t_sync) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    if dist is not None:
        # the ONLY collective: final accumulation of the row-sharded framebuffers (exact: the rows
        # a rank does not own are zero)
        dist.reduce(fb_tensor, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    st = dev.stats()
    rays_local = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    samples_local = st["camera_rays"]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rays_local, samples_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays_total, samples_total = float(c[0].item()), float(c[1].item())
    else:
        rays_total, samples_total = float(rays_local), float(samples_local)

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (closest-hit traversal), rank 0's launches
        launches = max(1, st["traverse_primary_launches"])
        avg_ms = st["ms_traverse_primary"] / launches
        # units per launch: replay the same steps with the work counters on (deterministic workload)
        cdev = Device(local_rank, acquire_stats=2, stream_capacity=CAPACITY)
        cdev.assign_scene(scene)
        cdev.resize(W, H)
        run(cdev, args.steps)
        cs = cdev.stats()
        cdev.close()
        n_primary = cs["camera_rays"] + cs["bounce_rays"]
        a_bytes = algorithmic_bytes(n_primary, cs["nodes_primary"], cs["tris_primary"], cs["leaves_primary"])
        a_per_launch = a_bytes / max(1, cs["traverse_primary_launches"])
        achieved = a_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # measured HBM-side bytes per launch of the same kernel: PMC passes of this command, summarised into
        # profiles/ by tools/prof_summary.py (counters cannot be read from inside the process)
        traffic, traffic_src, limiter = None, None, None
        tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if os.path.exists(tpath) and (W, H, spi) == (1920, 1080, SPI) and world == 1 and args.scene == SCENE:
            tk = json.load(open(tpath))["kernels"].get("k_traverse<false, false, false>")
            if tk:
                traffic, traffic_src = int(tk["hbm_bytes"]), f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, own passes, x2 read correction)"
                if "valu_lane_utilisation" in tk:
                    # what actually limits the kernel on this 40 KB scene (SURVEY.md 8d asks for it next to the HBM fraction)
                    limiter = {"kind": "VALU issue + latency (geometry is cache resident)", "valu_lane_utilisation": tk["valu_lane_utilisation"],
                               "wave_wait_share": tk.get("wave_wait_share"), "wave_issue_share": tk.get("wave_issue_share"), "source": f"profiles/{TRAFFIC_FILE} (SQ counters)"}
        roofline = {"bound": "hbm", "kernel": "k_traverse<closest>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "note": "achieved = SURVEY 8(d) algorithmic bytes (60 B/ray + 256 B/Node8 + 52 B/triangle + 96 B/entity leaf visited) / launch time; "
                            "the geometry term is served by L1/L2 on this 40 KB scene, so the figure can exceed what HBM delivers: `traffic` is "
                            "the measured HBM-side bytes per launch and `limiter` what actually bounds the kernel",
                    "limiter": limiter, "avg_launch_ms": round(avg_ms, 5), "launches": int(launches),
                    "avg_launch_ms_incl_warmup": round((st["ms_traverse_primary"] + warm["ms_traverse_primary"])
                                                       / max(1, launches + warm["traverse_primary_launches"]), 5),
                    "algorithmic_bytes_per_launch": int(a_per_launch)}

        stage_ms = {k: round(st[k], 3) for k in ("ms_generate", "ms_traverse_primary", "ms_shade", "ms_traverse_secondary", "ms_tail", "ms_resolve")}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            # bounded sample of the same workload: whole iterations until ~12 s of wall clock (at most 8)
            cw, ch = W, H
            cpu_rays = cpu_samples = 0
            n_it, threads = 0, 1
            t1 = time.perf_counter()
            while n_it < 8 and (n_it == 0 or time.perf_counter() - t1 < 12.0):
                _, os_ = oracle.render(scene, spi, cw, ch, iteration=n_it, seed=SEED)
                cpu_rays += os_["camera_rays"] + os_["bounce_rays"] + os_["shadow_rays"]
                cpu_samples += os_["camera_rays"]
                threads = int(os_["threads_used"])
                n_it += 1
            dt = time.perf_counter() - t1
            cpu = {"value": round(cpu_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"{n_it} iteration(s) of {os.path.basename(args.scene)} {cw}x{ch} spi {spi} (oracle/, CPU restatement of cpu_trace, not the AnyDSL binary)",
                   "msamples_per_s": round(cpu_samples / dt / 1e6, 3), "seconds": round(dt, 2)}

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
            "config": {"workload": f"{os.path.relpath(args.scene, ROOT)} {W}x{H}, path integrator, spi {spi} x {args.steps * (1 if by_rows else world)} iterations, seed {SEED}",
                       "sharding": "whole film" if shards == 1 else (f"film rows interleaved over {shards} GPUs + one RCCL reduce" if by_rows else
                                                                    f"{args.steps} full-film iterations per GPU (rank r: iterations r*K .. r*K+K-1) + one RCCL reduce")},
            "msamples_per_s": round(samples_total / elapsed / 1e6, 3),
            "rays": {"camera": st["camera_rays"], "bounce": st["bounce_rays"], "shadow": st["shadow_rays"], "scope": "rank 0"},
            "stage_ms_rank0": stage_ms,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }

    dev.close()
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
