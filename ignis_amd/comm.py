"""The launcher's half of a tile-sharded render: hand every rank the 128-byte RCCL id rank 0 made, then let the device library
talk RCCL itself (include/igd_device.h igd_comm_*, csrc/device/comm.hip). No torch, no MPI: one TCP rendezvous on
MASTER_ADDR : MASTER_PORT + 1 — the variables `python -m torch.distributed.run` and ignis_amd.cli's own launcher export.

    comm = Comm.agreed(dev)            # every rank, before rendering; None = "all ranks fall back" (never a hang, never a split)
    ... dev.render(..., row_offset=comm.rank, row_stride=comm.world) ...
    comm.gather_rows(dst=0)            # the path's one collective: each rank's rows into rank 0's framebuffer
    total = comm.allreduce([rays], "sum")

Bring-up is a vote, because ncclCommInitRank has no timeout and a collective one rank abandons hangs the others:

  1. every rank says over the rendezvous socket whether it can take part at all (librccl loaded, the id made) — BEFORE anyone
     enters ncclCommInitRank. Rank 0 answers "go" + the id only when all world - 1 peers have arrived and all said yes; otherwise
     "fall back" to whoever did arrive (a rank that never shows up times out on its own and falls back too);
  2. after "go" every rank brings its communicator up and runs one all-reduce on a helper thread under a deadline, reports success or
     failure over the same socket, and rank 0 broadcasts the verdict: "native" only if every rank succeeded in time. A rank left
     inside ncclCommInitRank by a peer that failed gets "fall back" from its main thread's deadline and abandons the helper thread
     (callers re-execute the process on the torch.distributed route, which also disposes of that thread).

The rendezvous recognises its own job: the hello carries a token of (MASTER_ADDR, MASTER_PORT, world, the launcher's run id), rank 0
acknowledges a valid hello at once (a peer waits 2 s for that before it tries the next candidate port), and the server binds to
MASTER_ADDR, not to every interface.
"""
import ctypes as C
import hashlib
import os
import socket
import struct
import threading
import time

from . import device as _device

ID_BYTES = 128


_MAGIC = b"IGD-RCCL-ID\x02"
_PORT_SPAN = 8  # rank 0 listens on the first free port of MASTER_PORT + 1 .. + 8; the others try them in turn
_TOKEN_BYTES = 16
_HELLO = struct.Struct("<ii16sB")  # rank, world, job token, "I can take part"
_ACK, _GO, _FALLBACK, _NATIVE, _ABORT = b"A", b"G", b"F", b"N", b"X"
ABORT = "abort: "  # prefix of a reason that no fallback can cure (ranks of the job are missing): callers give up instead of starting over


def job_token(world, addr=None, port=None):
    """What tells this job's ranks from another job's on the same host: the rendezvous address both were given, the world size and the
    launcher's run id (torch.distributed.run exports TORCHELASTIC_RUN_ID, ignis_amd.cli's launcher IGNIS_JOB_TOKEN)."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else os.environ.get("MASTER_PORT", "29511"))
    text = "|".join([addr, str(port), str(int(world)), os.environ.get("TORCHELASTIC_RUN_ID", ""), os.environ.get("IGNIS_JOB_TOKEN", "")])
    return hashlib.sha256(text.encode()).digest()[:_TOKEN_BYTES]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        part = conn.recv(n - len(buf))
        if not part:
            raise ConnectionError("peer closed the connection")
        buf += part
    return buf


def _listen(addr, first_port, backlog):
    last = None
    for p in range(first_port, first_port + _PORT_SPAN):
        for host in (addr, ""):  # MASTER_ADDR first; every interface only if that name is not an address of this host
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((host, p))
                srv.listen(backlog)
                return srv
            except OSError as e:
                srv.close()
                last = e
                if e.errno in (98, 48):  # EADDRINUSE: the next port, not the next host
                    break
    raise OSError(f"rank 0: no free port in {first_port} .. {first_port + _PORT_SPAN - 1} for the RCCL id exchange ({last})")


def _serve(srv, world, token, t_end):
    """Rank 0: accept until the world - 1 peers of this job have said hello (each acknowledged at once) or the deadline passes.
    Returns {peer rank: (connection, its vote)}."""
    peers = {}
    while len(peers) < world - 1:
        left = t_end - time.monotonic()
        if left <= 0:
            break
        srv.settimeout(left)
        try:
            conn, _ = srv.accept()
        except (socket.timeout, OSError):
            break
        try:
            conn.settimeout(2.0)
            hello = _recv_exact(conn, len(_MAGIC) + _HELLO.size)
            peer, peer_world, peer_token, vote = _HELLO.unpack(hello[len(_MAGIC):])
            if hello[:len(_MAGIC)] != _MAGIC or peer_token != token or peer_world != world or not (0 < peer < world) or peer in peers:
                conn.close()  # (a stray connection or another job's rank: it gets nothing and moves on to the next port)
                continue
            conn.sendall(_MAGIC + _ACK)
            peers[peer] = (conn, bool(vote))
        except (OSError, ConnectionError, struct.error):
            conn.close()
    return peers


def _join(rank, world, token, vote, addr, first_port, t_end):
    """Rank > 0: connect to this job's rank 0 (the candidate ports in turn until one acknowledges the hello) and return the socket."""
    while True:
        for p in range(first_port, first_port + _PORT_SPAN):
            try:
                conn = socket.create_connection((addr, p), timeout=2.0)
            except (OSError, ConnectionError):
                continue
            try:
                conn.settimeout(2.0)  # (a short wait for the acknowledgement: a foreign listener costs two seconds, not the job's deadline)
                conn.sendall(_MAGIC + _HELLO.pack(rank, world, token, 1 if vote else 0))
                if _recv_exact(conn, len(_MAGIC) + 1) == _MAGIC + _ACK:
                    return conn
            except (OSError, ConnectionError):
                pass
            conn.close()
        if time.monotonic() > t_end:
            raise TimeoutError(f"rank {rank}: no rank 0 of this job at {addr}:{first_port}..{first_port + _PORT_SPAN - 1}")
        time.sleep(0.05)


def _within(fn, t_end):
    """fn() on a helper thread; (True, result) if it returns before the deadline, (False, reason) if it raises or is still running
    (the thread is a daemon and is abandoned: only a new process image gets rid of a call stuck inside RCCL)."""
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:  # noqa: BLE001 - reported, not swallowed
            box["error"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(max(0.0, t_end - time.monotonic()))
    if th.is_alive():
        return False, "still inside the bring-up at the deadline"
    if "error" in box:
        return False, box["error"]
    return True, box.get("value")


def agree(rank, world, make_id, bring_up, probe=None, addr=None, port=None, deadline=60.0, arrive=None):
    """The two-phase bring-up vote (module docstring). make_id() -> the id bytes (rank 0 only); probe() raises if this rank cannot
    take part (ranks > 0); bring_up(id) creates the communicator and proves that it talks (may block: run under the deadline).
    Returns (True, "") on every rank or (False, reason) on every rank — never a mixture — within about two deadlines. A reason that starts
    with ABORT means that ranks of the job never showed up: no fallback collective can work either, callers exit instead of starting over."""
    # `deadline` bounds the bring-up (phase 2); the ranks get `arrive` seconds (default five deadlines) to show up at all: they reach this point
    # after loading a scene and allocating their streams, which on a fresh box differs by more than a collective may wait
    arrive = 5.0 * deadline if arrive is None else float(arrive)
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29511")) + 1)
    token = job_token(world, addr, base - 1)
    if world == 1:
        ok, why = _within(lambda: bring_up(make_id()), time.monotonic() + deadline)
        return (True, "") if ok else (False, str(why))
    if rank == 0:
        vote, why, blob = True, "", bytes(ID_BYTES)
        try:
            blob = make_id()
        except Exception as e:  # noqa: BLE001
            vote, why = False, f"rank 0 cannot take part ({type(e).__name__}: {e})"
        srv = _listen(addr, base, world)
        peers = {}
        try:
            peers = _serve(srv, world, token, time.monotonic() + arrive)
            missing = len(peers) < world - 1
            if missing:
                vote, why = False, f"{ABORT}only {len(peers) + 1} of {world} ranks arrived within {arrive:.0f} s"
            no = sorted(r for r, (_, v) in peers.items() if not v)
            if no:
                vote, why = False, why or f"rank(s) {no} cannot take part"
            for conn, _ in peers.values():
                try:
                    conn.sendall((_GO if vote else (_ABORT if missing else _FALLBACK)) + blob)
                except (OSError, ConnectionError):
                    pass  # (that peer reports nothing in phase 2: the verdict below becomes "fall back")
            if not vote:
                return False, why
            t_end = time.monotonic() + deadline
            ok, res = _within(lambda: bring_up(blob), t_end)
            why = "" if ok else f"rank 0: {res}"
            for r, (conn, _) in sorted(peers.items()):
                try:
                    conn.settimeout(max(0.1, t_end + 5.0 - time.monotonic()))
                    if _recv_exact(conn, 1) != b"\x01":
                        ok, why = False, why or f"rank {r} failed to bring its communicator up"
                except (OSError, ConnectionError):
                    ok, why = False, why or f"rank {r} did not report within the deadline"
            for conn, _ in peers.values():
                try:
                    conn.sendall(_NATIVE if ok else _FALLBACK)
                except (OSError, ConnectionError):
                    pass
            return (True, "") if ok else (False, why)
        finally:
            for conn, _ in peers.values():
                conn.close()
            srv.close()
    vote, why = True, ""
    if probe is not None:
        try:
            probe()
        except Exception as e:  # noqa: BLE001
            vote, why = False, f"rank {rank} cannot take part ({type(e).__name__}: {e})"
    try:
        conn = _join(rank, world, token, vote, addr, base, time.monotonic() + arrive)
    except TimeoutError as e:
        return False, ABORT + str(e)
    with conn:
        try:
            conn.settimeout(arrive + 10.0)  # rank 0 answers when everybody has arrived, or at its deadline
            reply = _recv_exact(conn, 1 + ID_BYTES)
        except (OSError, ConnectionError) as e:
            return False, f"rank {rank}: no decision from rank 0 ({type(e).__name__})"
        if reply[:1] == _ABORT:
            return False, f"{ABORT}rank 0 said: ranks of this job are missing"
        if reply[:1] != _GO or not vote:
            return False, why or "rank 0 said: fall back (a rank cannot take part)"
        t_end = time.monotonic() + deadline
        ok, res = _within(lambda: bring_up(reply[1:]), t_end)
        try:
            conn.sendall(b"\x01" if ok else b"\x00")
            conn.settimeout(max(0.1, t_end + 15.0 - time.monotonic()))
            final = _recv_exact(conn, 1)
        except (OSError, ConnectionError) as e:
            return False, f"rank {rank}: no verdict from rank 0 ({type(e).__name__})"
        if final == _NATIVE and ok:
            return True, ""
        return False, ("" if ok else f"rank {rank}: {res}") or "rank 0 said: fall back (another rank failed to bring its communicator up)"


def exchange_id(rank, world, make_id, addr=None, port=None, timeout=600.0):
    """Phase 1 on its own: rank 0 calls make_id() -> bytes and serves them to the world - 1 others; returns the id on every rank.
    `port` is the first candidate (default MASTER_PORT + 1: MASTER_PORT itself belongs to the launcher's store)."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29511")) + 1)
    token = job_token(world, addr, base - 1)
    if world == 1:
        return make_id()
    if rank == 0:
        blob = make_id()
        srv = _listen(addr, base, world)
        peers = {}
        try:
            peers = _serve(srv, world, token, time.monotonic() + timeout)
            if len(peers) < world - 1:
                raise TimeoutError(f"rank 0: only {len(peers) + 1} of {world} ranks arrived for the RCCL id exchange")
            for conn, _ in peers.values():
                conn.sendall(_GO + blob)
        finally:
            for conn, _ in peers.values():
                conn.close()
            srv.close()
        return blob
    with _join(rank, world, token, True, addr, base, time.monotonic() + timeout) as conn:
        conn.settimeout(timeout)
        reply = _recv_exact(conn, 1 + ID_BYTES)
    if reply[:1] != _GO:
        raise ConnectionError(f"rank {rank}: rank 0 did not hand out an id")
    return reply[1:]


class Comm:
    """The RCCL communicator a Device owns (igd_comm_init); rank / world as the launcher numbered the processes."""

    def __init__(self, dev, rank, world, addr=None, port=None, blob=None):
        self.dev, self.rank, self.world = dev, int(rank), int(world)
        if blob is None:
            blob = exchange_id(self.rank, self.world, self._make_id, addr, port)
        _device._check(_device.lib().igd_comm_init(dev._h, (C.c_uint8 * ID_BYTES).from_buffer_copy(blob), self.rank, self.world))

    @staticmethod
    def _make_id():
        buf = (C.c_uint8 * ID_BYTES)()
        _device._check(_device.lib().igd_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def _probe():
        if not _device.lib().igd_comm_available():
            raise RuntimeError(_device.lib().igd_last_error().decode() or "librccl.so is not available")

    @classmethod
    def agreed(cls, dev, rank=None, world=None, addr=None, port=None, deadline=60.0, fail=None):
        """The communicator of every rank, or None on every rank (module docstring) — call it BEFORE rendering: a caller that gets
        None still has all its work in front of it and takes the torch.distributed route. `fail`: tests — this rank's bring-up
        raises ("raise"), never returns ("hang"), or its librccl "does not load" ("probe")."""
        rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        made = {}

        def bring_up(blob):
            if fail == "raise":
                raise RuntimeError("injected bring-up failure")
            if fail == "hang":
                time.sleep(3600)
            c = cls(dev, rank, world, blob=blob)
            made["comm"] = c
            c.barrier()  # (the first collective: a communicator that cannot talk fails or stalls here, inside the deadline)
            return True

        def probe():
            if fail == "probe":
                raise RuntimeError("injected: librccl.so does not load")
            cls._probe()
        ok, why = agree(rank, world, cls._make_id, bring_up, probe if rank else None, addr, port, deadline)
        if ok:
            return made["comm"]
        cls.last_fallback_reason = why
        return None

    last_fallback_reason = ""

    @classmethod
    def from_env(cls, dev):
        return cls(dev, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))

    def world_size_from_backend(self):
        """ncclCommCount: what RCCL itself says the communicator spans."""
        return int(_device.lib().igd_comm_world_size(self.dev._h))

    def gather_rows(self, dst=0):
        _device._check(_device.lib().igd_comm_gather_rows(self.dev._h, dst))

    def allreduce(self, values, op="sum"):
        arr = (C.c_double * len(values))(*[float(v) for v in values])
        _device._check(_device.lib().igd_comm_allreduce_f64(self.dev._h, arr, len(values), {"sum": 0, "max": 2}[op]))
        return list(arr)

    def barrier(self):
        self.allreduce([0.0], "sum")

    def close(self):
        if self.dev is not None and getattr(self.dev, "_h", None):
            _device.lib().igd_comm_destroy(self.dev._h)
        self.dev = None
