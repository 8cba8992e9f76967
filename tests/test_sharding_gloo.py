"""N > 1 path on CPU: world_size-2 gloo processes shard the film by interleaved rows, render their
shard (with the oracle standing in for the device — this is a test of the partition + collective,
not of the kernels) and reduce to rank 0; the result must be the single-process image."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES

W, H, SPI, SEED = 48, 40, 2, 9


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, collective="reduce"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import oracle
    from ignis_amd import sharding
    from ignis_amd.tables import LoadedScene

    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = LoadedScene.from_file(os.path.join(SCENES, "diamond_scene.json"), W, H)
    fb, st = oracle.render(scene, SPI, W, H, seed=SEED, threads=2, rows=sharding.shard_settings(rank, world))
    assert sharding.check_shard(fb, rank, world)
    assert st["camera_rays"] == len(sharding.shard_rows(rank, world, H)) * W * SPI
    t = torch.from_numpy(fb)
    if collective == "gather":
        sharding.gather_rows(t, rank, world, dist, dst=0)  # bench.py's collective: owned rows only
    else:
        sharding.reduce_framebuffer(t, dist, dst=0)
    rays = torch.tensor([st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]], dtype=torch.float64)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(out_path, fb=t.numpy(), rays=rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_iterations(rank, world, port, out_path, steps):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import oracle
    from ignis_amd import sharding
    from ignis_amd.tables import LoadedScene

    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = LoadedScene.from_file(os.path.join(SCENES, "diamond_scene.json"), W, H)
    fb = np.zeros((H, W, 3), np.float32)
    rays = 0
    for it in sharding.shard_iterations(rank, world, steps):
        _, st = oracle.render(scene, SPI, W, H, iteration=it, seed=SEED, threads=2, fb=fb)
        rays += st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    t = torch.from_numpy(fb)
    sharding.reduce_framebuffer(t, dist, dst=0)
    r = torch.tensor([rays], dtype=torch.float64)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(out_path, fb=t.numpy(), rays=r.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_iteration_sharding_sums_to_the_single_process_image(tmp_path):
    """bench.py --sharding iterations: rank r renders iterations r*K .. r*K+K-1; reduce(SUM) = all iterations."""
    import torch.multiprocessing as mp

    import oracle
    from ignis_amd.tables import LoadedScene

    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker_iterations, args=(2, _free_port(), out, 2), nprocs=2, join=True)
    got = np.load(out)
    scene = LoadedScene.from_file(os.path.join(SCENES, "diamond_scene.json"), W, H)
    ref = np.zeros((H, W, 3), np.float32)
    rays = 0
    for it in range(4):
        _, st = oracle.render(scene, SPI, W, H, iteration=it, seed=SEED, threads=2, fb=ref)
        rays += st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    np.testing.assert_allclose(got["fb"], ref, rtol=2e-5, atol=1e-6)  # (i0 + i1) + (i2 + i3) vs ((i0 + i1) + i2) + i3
    assert int(got["rays"][0]) == rays


@pytest.mark.parametrize("collective,world", [("reduce", 2), ("gather", 2), ("gather", 3)])
def test_two_rank_row_sharding_reassembles_the_image(tmp_path, collective, world):
    """bench.py's default N > 1 partition (tile-sharded film, gather of the owned rows; H = 40 is not a multiple of 3, so the
    three-rank case has ragged shards)."""
    import torch.multiprocessing as mp

    import oracle
    from ignis_amd.tables import LoadedScene

    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(world, _free_port(), out, collective), nprocs=world, join=True)
    got = np.load(out)

    scene = LoadedScene.from_file(os.path.join(SCENES, "diamond_scene.json"), W, H)
    ref, st = oracle.render(scene, SPI, W, H, seed=SEED, threads=2)
    # per-pixel results are independent of the tiling; only the float summation order inside a pixel differs
    np.testing.assert_allclose(got["fb"], ref, rtol=2e-5, atol=1e-6)
    assert int(got["rays"][0]) == st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]


def test_shard_rows_partition():
    from ignis_amd import sharding
    for world in (1, 2, 3, 8):
        rows = np.concatenate([sharding.shard_rows(r, world, 37) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(37))
    with pytest.raises(ValueError):
        sharding.shard_rows(2, 2, 10)


class _OracleRuntime:
    """What ignis_amd.cli needs of a Runtime, with the oracle standing in for the device: the test is about the CLI's sharding,
    its one collective and what rank 0 writes — not about the kernels (no GPU here)."""

    def __init__(self, path, opts):
        import oracle
        from ignis_amd.tables import LoadedScene
        self._oracle = oracle
        w, h = opts.OverrideFilmSize
        self._scene = LoadedScene.from_file(path, w, h)
        self._opts = opts
        self.FramebufferWidth, self.FramebufferHeight = w, h
        self.SPI = opts.SPI
        self.IterationCount = self.SampleCount = 0
        self._fb = np.zeros((h, w, 3), np.float32)
        self._stats = {"camera_rays": 0, "bounce_rays": 0, "shadow_rays": 0}
        self.device = None  # (no igd_device behind it: a native communicator cannot come up on this "runtime")

    def recommendedBatch(self):
        return 2

    def stepMany(self, count):
        for _ in range(count):
            _, st = self._oracle.render(self._scene, self.SPI, self.FramebufferWidth, self.FramebufferHeight, iteration=self.IterationCount,
                                        seed=self._opts.Seed, threads=2, fb=self._fb, rows=(self._opts.RowOffset, self._opts.RowStride))
            for k in self._stats:
                self._stats[k] += st[k]
            self.IterationCount += 1
            self.SampleCount += self.SPI

    def synchronize(self):
        pass

    def framebufferTensor(self, torch, on_device=True):
        assert not on_device
        return torch.from_numpy(self._fb)

    def getStatistics(self):
        return dict(self._stats)

    def saveFramebuffer(self, path, fb=None):
        from ignis_amd.tables import save_exr
        save_exr(path, self._fb if fb is None else np.asarray(fb, np.float32), 1.0 / max(1, self.IterationCount), {"igSPP": self.SampleCount})
        return True

    def shutdown(self):
        self._scene.close()


def _cli_rank(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ignis_amd import cli
    rc = cli.main([os.path.join(SCENES, "diamond_scene.json"), "--spp", str(3 * SPI), "--spi", str(SPI), "--width", str(W), "--height", str(H),
                   "--seed", str(SEED), "-o", out_path, "--gpus", str(world), "--backend", "gloo", "--stats"], load=_OracleRuntime)
    assert rc == 0


@pytest.mark.parametrize("world", [2, 3])
def test_cli_gpus_n_entry_point_over_gloo(tmp_path, world):
    """`python -m ignis_amd.cli --gpus N` (the product entry point of the tile-sharded path): every rank runs cli.main with its
    RANK / WORLD_SIZE, renders the rows it owns, ONE gather of the owned rows, rank 0 alone writes the EXR = mean over the
    iterations of the single-process image."""
    import torch.multiprocessing as mp

    import oracle
    from ignis_amd.tables import LoadedScene
    from test_abi import _read_exr

    out = str(tmp_path / "sharded.exr")
    mp.spawn(_cli_rank, args=(world, _free_port(), out), nprocs=world, join=True)
    planes, attrs = _read_exr(out)
    got = np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    scene = LoadedScene.from_file(os.path.join(SCENES, "diamond_scene.json"), W, H)
    ref = np.zeros((H, W, 3), np.float32)
    for it in range(3):
        oracle.render(scene, SPI, W, H, iteration=it, seed=SEED, threads=2, fb=ref)
    np.testing.assert_allclose(got, ref / np.float32(3), rtol=2e-5, atol=1e-6)
    assert attrs["igSPP"][1] == str(3 * SPI).encode()


def _id_rank(rank, world, port, out_dir):
    from ignis_amd.comm import exchange_id, ID_BYTES
    blob = exchange_id(rank, world, lambda: bytes(range(ID_BYTES)), addr="127.0.0.1", port=port, timeout=60)
    with open(os.path.join(out_dir, f"id{rank}.bin"), "wb") as f:
        f.write(blob)


@pytest.mark.parametrize("world,first_port_taken", [(1, False), (3, False), (3, True)])
def test_rccl_id_rendezvous_without_torch(tmp_path, world, first_port_taken):
    """The launcher's half of the native RCCL path (ignis_amd/comm.py): rank 0's 128-byte id reaches every rank over one TCP
    exchange, whatever order the ranks come up in (no torch.distributed, no store) — also when something else already listens on the
    first candidate port (rank 0 moves to the next, the others recognise it by the magic word)."""
    import multiprocessing as mp
    import socket
    port = _free_port()
    squatter = None
    if first_port_taken:
        squatter = socket.socket()
        squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        squatter.bind(("127.0.0.1", port))
        squatter.listen(8)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_id_rank, args=(r, world, port, str(tmp_path))) for r in reversed(range(world))]  # (rank 0 last)
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    if squatter is not None:
        squatter.close()
    for r in range(world):
        assert (tmp_path / f"id{r}.bin").read_bytes() == bytes(range(128))


# ---- the bring-up vote of the native RCCL path (ignis_amd/comm.py agree): all ranks native or all ranks on the
# fallback, within the deadline, whatever one of them does. The communicator itself is faked (no GPU here): the protocol is the subject.

def _vote_rank(rank, world, port, out_dir, mode, bad_rank, deadline, token):
    import time
    sys.path.insert(0, ROOT)
    os.environ["IGNIS_JOB_TOKEN"] = token
    from ignis_amd.comm import ID_BYTES, agree
    if mode == "absent" and rank == bad_rank:
        return  # this rank dies before it says hello
    marker = os.path.join(out_dir, f"entered{rank}")

    def bring_up(blob):
        open(marker, "w").close()
        assert blob == bytes(range(ID_BYTES))
        if rank == bad_rank and mode == "raise":
            raise RuntimeError("ncclCommInitRank failed (injected)")
        if rank == bad_rank and mode == "hang":
            time.sleep(3600)
        if mode in ("raise", "hang"):
            time.sleep(3600 if rank != bad_rank else 0)  # the healthy ranks sit in a collective the failed one never joins
        return True

    def probe():
        if rank == bad_rank and mode == "probe":
            raise RuntimeError("librccl.so could not be loaded (injected)")
    t0 = time.monotonic()
    ok, why = agree(rank, world, lambda: bytes(range(ID_BYTES)), bring_up, probe if rank else None, addr="127.0.0.1", port=port, deadline=deadline)
    with open(os.path.join(out_dir, f"verdict{rank}"), "w") as f:
        f.write(f"{int(ok)} {time.monotonic() - t0:.2f} {why}")
    os._exit(0)  # (like the callers' re-exec: a helper thread may still be asleep inside the fake collective)


def _run_vote(tmp_path, world, mode, bad_rank, deadline, port=None, token="t"):
    import multiprocessing as mp
    port = port or _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_vote_rank, args=(r, world, port, str(tmp_path), mode, bad_rank, deadline, token)) for r in reversed(range(world))]
    for p in procs:
        p.start()
    return procs


def _verdicts(tmp_path, procs, world, absent=()):
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    out = {}
    for r in range(world):
        if r in absent:
            continue
        ok, secs, *why = (tmp_path / f"verdict{r}").read_text().split(" ", 2)
        out[r] = (bool(int(ok)), float(secs), why[0] if why else "")
    return out


def test_bring_up_vote_all_ranks_native(tmp_path):
    v = _verdicts(tmp_path, _run_vote(tmp_path, 3, "fine", -1, 30.0), 3)
    assert all(ok for ok, _, _ in v.values()) and all(secs < 20 for _, secs, _ in v.values())
    assert all((tmp_path / f"entered{r}").exists() for r in range(3))


@pytest.mark.parametrize("mode,bad_rank", [("raise", 1), ("raise", 0), ("hang", 2), ("hang", 0)])
def test_bring_up_vote_one_failing_rank_sends_every_rank_to_the_fallback(tmp_path, mode, bad_rank):
    """One of three ranks fails inside its bring-up — raises, or never returns — while the other two wait in a collective it never joins:
    all three end up on the fallback, the waiting ones released by the deadline (4 s here) instead of hanging."""
    v = _verdicts(tmp_path, _run_vote(tmp_path, 3, mode, bad_rank, 4.0), 3)
    assert not any(ok for ok, _, _ in v.values()), v
    assert all(secs < 4.0 * 2 + 25 for _, secs, _ in v.values()), v
    assert any("rank" in why for _, _, why in v.values())


def test_bring_up_vote_a_rank_without_rccl_keeps_everyone_out_of_the_collective(tmp_path):
    """Phase 1: a rank whose librccl does not load says so BEFORE anyone enters ncclCommInitRank — nobody's bring-up runs."""
    v = _verdicts(tmp_path, _run_vote(tmp_path, 3, "probe", 2, 20.0), 3)
    assert not any(ok for ok, _, _ in v.values())
    assert all(secs < 15 for _, secs, _ in v.values())
    assert not any((tmp_path / f"entered{r}").exists() for r in range(3))
    assert "cannot take part" in v[0][2]


def test_bring_up_vote_a_rank_that_never_arrives(tmp_path):
    v = _verdicts(tmp_path, _run_vote(tmp_path, 3, "absent", 1, 3.0), 3, absent=(1,))
    assert not any(ok for ok, _, _ in v.values())
    assert all(secs < 3.0 * 2 + 25 for _, secs, _ in v.values())
    assert "2 of 3 ranks arrived" in v[0][2] and all(why.startswith("abort: ") for _, _, why in v.values())  # no fallback can cure a missing rank: callers give up
    assert not any((tmp_path / f"entered{r}").exists() for r in (0, 2))


def test_two_jobs_on_adjacent_ports_do_not_cross_connect(tmp_path):
    """two jobs of the same world size whose MASTER_PORTs are neighbours share seven of their eight candidate ports. The
    hello's job token keeps a rank of one from being served by the other's rank 0: both jobs come up, each with its own ranks."""
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(), b.mkdir()
    port = _free_port()
    pa = _run_vote(a, 3, "fine", -1, 30.0, port=port, token="job-a")
    pb = _run_vote(b, 3, "fine", -1, 30.0, port=port + 1, token="job-b")
    va, vb = _verdicts(a, pa, 3), _verdicts(b, pb, 3)
    assert all(ok for ok, _, _ in va.values()) and all(ok for ok, _, _ in vb.values())


def _cli_rank_native_then_fallback(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      IGNIS_CLI_FALLBACK_BACKEND="gloo", IGNIS_COMM_DEADLINE="20")
    from ignis_amd import cli
    seen = []

    def reexec(argv):  # (the product replaces the process image; here the same interpreter runs the new command line)
        seen.append(list(argv))
        return cli.main(argv, load=_OracleRuntime, reexec=reexec)
    rc = cli.main([os.path.join(SCENES, "diamond_scene.json"), "--spp", str(2 * SPI), "--spi", str(SPI), "--width", str(W), "--height", str(H),
                   "--seed", str(SEED), "-o", out_path, "--gpus", str(world), "--backend", "rccl"], load=_OracleRuntime, reexec=reexec)
    assert rc == 0
    assert len(seen) == 1 and seen[0][-2:] == ["--backend", "gloo"] and seen[0].count("--backend") == 1


def test_cli_native_backend_falls_back_on_every_rank_when_rccl_cannot_come_up(tmp_path):
    """`python -m ignis_amd.cli --gpus 3` with its default backend on a box without GPUs: rank 0 cannot make an RCCL id, says so in the
    bring-up vote (ignis_amd/comm.py), and ALL three ranks start over with the fallback backend before anything was rendered — nobody
    waits in a collective, the finished image is the single-process one."""
    import torch.multiprocessing as mp

    import oracle
    from ignis_amd.tables import LoadedScene
    from test_abi import _read_exr

    out = str(tmp_path / "fallback.exr")
    mp.spawn(_cli_rank_native_then_fallback, args=(3, _free_port(), out), nprocs=3, join=True)
    planes, _ = _read_exr(out)
    got = np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    scene = LoadedScene.from_file(os.path.join(SCENES, "diamond_scene.json"), W, H)
    ref = np.zeros((H, W, 3), np.float32)
    for it in range(2):
        oracle.render(scene, SPI, W, H, iteration=it, seed=SEED, threads=2, fb=ref)
    np.testing.assert_allclose(got, ref / np.float32(2), rtol=2e-5, atol=1e-6)
