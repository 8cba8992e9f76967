"""igcli counterpart for the MI355X backend (src/frontend/cli/main.cpp:60-185): render a scene file for a number of
samples per pixel and write the EXR, printing the reference's statistics lines (ray counts, min/med/max Msamples/s).

    python -m ignis_amd.cli scenes/diamond_scene.json --spp 64 -o out.exr [--width W --height H --spi N --seed S --stats]
    python -m ignis_amd.cli scenes/diamond_scene.json --spp 1024 --gpus 8 -o out.exr

`--gpus N`: one process per GPU (spawned here, or by torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment). The film is tile-sharded — rank r renders rows r, r + N, ... of every iteration (igd_render_settings.row_offset /
row_stride), scene replicated, no data-path exchange — and the only collective is ONE gather of the owned rows to rank 0 over RCCL
(xGMI) when rendering is done (ignis_amd/sharding.py); rank 0 writes the EXR. The image is the single-GPU one bit for bit.
"""
import argparse
import datetime
import math
import os
import socket
import subprocess
import sys
import time

from .runtime import RuntimeOptions, loadFromFile


def beautiful_time(ms):
    s = ms / 1000.0
    return f"{s:.3f}s" if s < 60 else f"{int(s // 60)}m {s % 60:.1f}s"


def _spawn_ranks(argv, gpus):
    """The launcher half of --gpus N: N copies of this command, one per GPU, rendezvous on 127.0.0.1."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    nonce = os.urandom(8).hex()  # (ignis_amd/comm.py job_token: the ranks of this launch recognise each other by it)
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IGNIS_JOB_TOKEN=nonce)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-m", "ignis_amd.cli"] + list(argv), env=env))
    # all ranks are watched together: the first one that fails (a negative code = killed by a signal counts) takes the others down, which
    # would otherwise wait in the gather for a peer that is gone
    failed = 0
    while procs and not failed:
        for p in list(procs):
            rc = p.poll()
            if rc is not None:
                procs.remove(p)
                failed = failed or (1 if rc != 0 else 0)
        time.sleep(0.05)
    for p in procs:
        p.terminate()
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return failed


def _reexec(argv):
    """Replaces this process by a fresh `python -m ignis_amd.cli` (a rank that falls back to torch.distributed must import torch before
    any HIP library is loaded, and may have a helper thread inside ncclCommInitRank to get rid of)."""
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, [sys.executable, "-m", "ignis_amd.cli"] + list(argv))


def main(argv=None, load=loadFromFile, reexec=_reexec):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(prog="ignis_amd.cli", description=__doc__.splitlines()[0])
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default="output.exr")
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel (rounded up to a multiple of the spi)")
    ap.add_argument("--spi", type=int, default=0, help="samples per iteration, 0 = recommended")
    ap.add_argument("--time", type=float, default=None, help="render for this many seconds instead of a fixed spp")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="iterations rendered as one wavefront per call (0 = as many as fill the GPU, 1 = like the reference)")
    ap.add_argument("--stats", action="store_true", help="acquire ray statistics and stage timers")
    ap.add_argument("--full-stats", action="store_true", help="also count traversal work (slower kernels)")
    ap.add_argument("--gpus", type=int, default=1, help="tile-shard the film over this many GPUs of the node (one process each, one RCCL gather at the end)")
    ap.add_argument("--backend", default="rccl", choices=("rccl", "nccl", "gloo"),
                    help="--gpus N: rccl = the device library's own RCCL communicator (igd_comm_*, no torch); nccl / gloo = torch.distributed with that backend (gloo: CPU tests)")
    args = ap.parse_args(argv)
    if args.spp is None and args.time is None:
        args.spp = 64
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        return _spawn_ranks(argv, args.gpus)
    if world != max(1, args.gpus):
        print(f"--gpus {args.gpus} does not match WORLD_SIZE {world}", file=sys.stderr)
        return 2
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    native = args.backend == "rccl"
    sharded = world > 1 or bool(os.environ.get("IGNIS_CLI_FORCE_DIST"))  # (the variable: a single rank through the whole RCCL path, for tests)
    if sharded:
        if args.time is not None:
            print("--time is not available with --gpus N (ranks must render the same iterations)", file=sys.stderr)
            return 2
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not native:
            import torch  # first: its HIP runtime and RCCL are the ones the device library binds to in this process
            import torch.distributed as dist
            if args.backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=30))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
        args.gpu = local_rank
    chatty = rank == 0

    t_all = time.perf_counter()
    opts = RuntimeOptions.makeDefault()
    opts.Device = args.gpu
    opts.SPI = args.spi
    opts.Seed = args.seed
    opts.OverrideFilmSize = (args.width, args.height)
    opts.AcquireStats = True if args.full_stats else (1 if args.stats else False)
    if sharded:
        from . import sharding
        opts.RowOffset, opts.RowStride = sharding.shard_settings(rank, world)
    t0 = time.perf_counter()
    rt = load(args.scene, opts)
    t_loading = time.perf_counter() - t0

    comm = None
    if sharded and native:
        # The communicator FIRST: brought up by a vote among the ranks before anything is rendered (ignis_amd/comm.py agree), so
        # that a communicator that cannot come up costs nothing — the ranks then all start over on the torch.distributed route, none is
        # left waiting in a collective — and so that its set-up is not hidden behind the render.
        from .comm import Comm
        fail = os.environ.get("IGNIS_COMM_FAIL")  # tests: "raise" / "hang" / "probe" on rank IGNIS_COMM_FAIL_RANK (default: every rank)
        if fail and int(os.environ.get("IGNIS_COMM_FAIL_RANK", rank)) != rank:
            fail = None
        t0 = time.perf_counter()
        comm = Comm.agreed(rt.device, rank, world, deadline=float(os.environ.get("IGNIS_COMM_DEADLINE", "60")), fail=fail)
        t_comm = time.perf_counter() - t0
        if comm is None and Comm.last_fallback_reason.startswith("abort: "):
            print(f"rank {rank}: {Comm.last_fallback_reason} — the job is incomplete, giving up", file=sys.stderr, flush=True)
            return 3
        if comm is None:
            fallback = os.environ.get("IGNIS_CLI_FALLBACK_BACKEND", "nccl")
            print(f"rank {rank}: the ranks agreed not to use the native RCCL communicator ({Comm.last_fallback_reason}); starting over with --backend {fallback}", file=sys.stderr, flush=True)
            rest = [a for i, a in enumerate(argv) if not (a == "--backend" or (i > 0 and argv[i - 1] == "--backend") or a.startswith("--backend="))]
            os.environ.pop("IGNIS_COMM_FAIL", None)
            return reexec(rest + ["--backend", fallback])
        if chatty:
            print(f"RCCL communicator of {comm.world_size_from_backend()} ranks up in {beautiful_time(t_comm * 1e3)}", file=sys.stderr)

    spi = rt.SPI
    desired_iter = int(math.ceil((args.spp or 0) / spi))
    if args.spp and args.spp % spi:
        if chatty:
            print(f"Given spp {args.spp} is not a multiple of the spi {spi}. Using spp {desired_iter * spi} instead", file=sys.stderr)
    if chatty:
        print("Started rendering..." + (f" ({world} GPUs, film rows interleaved)" if world > 1 else ""), file=sys.stderr)
    samples_sec, t_render = [], 0.0
    batch = args.batch if args.batch > 0 else rt.recommendedBatch()
    while True:
        count = batch if desired_iter <= 0 else min(batch, desired_iter - rt.IterationCount)
        t0 = time.perf_counter()
        rt.stepMany(count)
        rt.synchronize()  # per-call timing like the reference's blocking step()
        dt = time.perf_counter() - t0
        t_render += dt
        samples_sec += [spi * rt.FramebufferWidth * rt.FramebufferHeight * count / dt] * count
        if desired_iter > 0 and rt.IterationCount >= desired_iter:
            break
        if args.time is not None and t_render > args.time:
            break

    gathered = None
    st = rt.getStatistics() if (args.stats or args.full_stats) else None
    if sharded and native:
        # the ONLY collective: the rows each rank owns, to rank 0 (W x H x 12 / world bytes per rank), by the device library itself
        t0 = time.perf_counter()
        comm.gather_rows(dst=0)
        t_render += time.perf_counter() - t0
        if rank == 0:
            gathered = rt.getFramebufferForHost().copy()
        if st is not None:
            keys = [k for k in ("camera_rays", "bounce_rays", "shadow_rays", "nodes", "tris", "leaves") if k in st]
            st = dict(st, **{k: int(v) for k, v in zip(keys, comm.allreduce([float(st[k]) for k in keys], "sum"))})
    elif sharded:
        from . import sharding
        t0 = time.perf_counter()
        fb = rt.framebufferTensor(torch, on_device=args.backend == "nccl")
        sharding.gather_rows(fb, rank, world, dist, dst=0)
        if args.backend == "nccl":
            torch.cuda.synchronize()
        t_render += time.perf_counter() - t0
        if rank == 0:
            gathered = fb.cpu().numpy()
        if st is not None:
            keys = [k for k in ("camera_rays", "bounce_rays", "shadow_rays", "nodes", "tris", "leaves") if k in st]
            c = torch.tensor([float(st[k]) for k in keys], dtype=torch.float64, device=fb.device)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            st = dict(st, **{k: int(v) for k, v in zip(keys, c.tolist())})
    ok = True
    t_saving = 0.0
    if rank == 0:
        t0 = time.perf_counter()
        ok = rt.saveFramebuffer(args.output, gathered)
        t_saving = time.perf_counter() - t0
        print(f"Result saved to {args.output}" if ok else f"Failed to save EXR file {args.output}", file=sys.stderr)
    ms_all = (time.perf_counter() - t_all) * 1e3
    if comm is not None:
        comm.barrier()
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        rt.shutdown()
        return 0
    if args.stats or args.full_stats:
        total = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
        print("Statistics:\n"
              f"  Ray Count: {total}\n    Camera: {st['camera_rays']}\n    Bounce: {st['bounce_rays']}\n    Shadow: {st['shadow_rays']}\n"
              f"  Mrays/s (render time): {total / t_render / 1e6:.1f}")
        if args.full_stats:
            print(f"  Traversal: nodes {st['nodes']}, triangles {st['tris']}, entity leaves {st['leaves']}")
    print(f"  Iterations: {rt.IterationCount}\n  SPP: {rt.SampleCount}\n  SPI: {spi}\n  Time: {beautiful_time(ms_all)}\n"
          f"    Loading> {beautiful_time(t_loading * 1e3)}\n    Render>  {beautiful_time(t_render * 1e3)}\n    Saving>  {beautiful_time(t_saving * 1e3)}")
    rt.shutdown()
    s = sorted(samples_sec)
    print(f"# {s[0] * 1e-6:.3f}/{s[len(s) // 2] * 1e-6:.3f}/{s[-1] * 1e-6:.3f} (min/med/max Msamples/s)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
