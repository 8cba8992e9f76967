"""The launcher's half of a tile-sharded render: hand every rank the 128-byte RCCL id rank 0 made, then let the device library
talk RCCL itself (include/igd_device.h igd_comm_*, csrc/device/comm.hip). No torch, no MPI: one TCP exchange on
MASTER_ADDR : MASTER_PORT + 1 — the variables `python -m torch.distributed.run` and ignis_amd.cli's own launcher export.

    comm = Comm.from_env(dev)          # RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    ... dev.render(..., row_offset=comm.rank, row_stride=comm.world) ...
    comm.gather_rows(dst=0)            # the path's one collective: each rank's rows into rank 0's framebuffer
    total = comm.allreduce([rays], "sum")
"""
import ctypes as C
import os
import socket
import struct
import time

from . import device as _device

ID_BYTES = 128


_MAGIC = b"IGD-RCCL-ID\x01"
_PORT_SPAN = 8  # rank 0 listens on the first free port of MASTER_PORT + 1 .. + 8; the others try them in turn


def exchange_id(rank, world, make_id, addr=None, port=None, timeout=600.0):
    """Rank 0 calls make_id() -> bytes and serves them to the world - 1 others; returns the id on every rank. `port` is the first
    candidate (default MASTER_PORT + 1: MASTER_PORT itself belongs to the launcher's store); a peer is recognised by a magic word,
    so a port that something else listens on is skipped."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29511")) + 1)
    if world == 1:
        return make_id()
    if rank == 0:
        blob = make_id()
        srv = None
        for p in range(port, port + _PORT_SPAN):
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind(("", p))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise OSError(f"rank 0: no free port in {port} .. {port + _PORT_SPAN - 1} for the RCCL id exchange")
        srv.listen(world)
        srv.settimeout(timeout)
        served = set()
        try:
            while len(served) < world - 1:
                conn, _ = srv.accept()
                with conn:
                    try:
                        conn.settimeout(10.0)
                        hello = _recv_exact(conn, len(_MAGIC) + 8)
                        peer, peer_world = struct.unpack("<ii", hello[len(_MAGIC):])
                        if hello[:len(_MAGIC)] != _MAGIC or peer_world != world or not (0 < peer < world):
                            continue  # (a stray connection: it gets nothing)
                        conn.sendall(_MAGIC + blob)
                        served.add(peer)
                    except (OSError, ConnectionError, struct.error):
                        continue
        finally:
            srv.close()
        return blob
    deadline = time.monotonic() + timeout
    while True:
        for p in range(port, port + _PORT_SPAN):
            try:
                with socket.create_connection((addr, p), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(_MAGIC + struct.pack("<ii", rank, world))
                    reply = _recv_exact(conn, len(_MAGIC) + ID_BYTES)
                if reply[:len(_MAGIC)] == _MAGIC:
                    return reply[len(_MAGIC):]
            except (OSError, ConnectionError):
                pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no RCCL id from rank 0 at {addr}:{port}..{port + _PORT_SPAN - 1}")
        time.sleep(0.05)


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        part = conn.recv(n - len(buf))
        if not part:
            raise ConnectionError("peer closed the connection")
        buf += part
    return buf


class Comm:
    """The RCCL communicator a Device owns (igd_comm_init); rank / world as the launcher numbered the processes."""

    def __init__(self, dev, rank, world, addr=None, port=None):
        self.dev, self.rank, self.world = dev, int(rank), int(world)
        lib = _device.lib()

        def make_id():
            buf = (C.c_uint8 * ID_BYTES)()
            _device._check(lib.igd_comm_unique_id(buf))
            return bytes(buf)
        blob = exchange_id(self.rank, self.world, make_id, addr, port)
        _device._check(lib.igd_comm_init(dev._h, (C.c_uint8 * ID_BYTES).from_buffer_copy(blob), self.rank, self.world))

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
