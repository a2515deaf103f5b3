"""Multi-GPU host logic: rebalancing windows shard embarrassingly across the GPUs of one node.

Windows are split into contiguous ranges (rank r gets `partition(W, world)[r]`), each range runs through its
own `_native.Device`, and ONE gather of the `[W_local x k]` weights goes to rank 0 over RCCL/xGMI.  There is
no other data-path collective.  Two ways to drive the GPUs:

* one process per GPU (`bench.py --gpus N`, `torch.distributed.run`-style environment): `ControlPlane` is
  the rendezvous - a few small host messages (the 128-byte RCCL id, barriers, timing reductions) over plain
  TCP sockets of the standard library.  Nothing here imports torch; a test may inject another transport
  (e.g. gloo) through `ControlPlane(transport=...)`;
* one process, all GPUs (`run_sharded` + `_native.DeviceGroup`): what `backtest_portfolio` uses, since the
  reference's driver is a single process (`/root/reference/src/main.py:26`); the communicator comes from
  `tp_comm_init_all`, no rendezvous at all.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import tempfile
import threading
import time

import numpy as np

_MAGIC = b"TPCP1\0\0\0"
_TIMEOUT_S = float(os.environ.get("TP_CONTROL_TIMEOUT", "180"))


def partition(W: int, world: int) -> list[tuple[int, int]]:
    """Contiguous, balanced window ranges [lo, hi) for ranks 0..world-1 (first W % world ranks get one more)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, extra = divmod(int(W), world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


# The register-tile kernels take the whole aligned 16-row (32-row) blocks of a window from block-window sums over the panel
# (csrc/posterior_fused.hip, tp_window_sums_kernel).  Every sum holds the blocks of ITS window only, but the order in
# which they are added depends on the block's position in the UPLOADED panel: windows sharded with the whole panel
# (`run_sharded`, `bench.py`: what the product does) are bit-identical to the unsharded run; a shard that uploads only a
# slice of the panel (this helper: tests and callers short of memory) cuts it at a multiple of 512 rows so that the 16- and
# 32-row blocks stay the same, and agrees with the full-panel run to rounding (a few ulps of S1), not bit for bit.
PANEL_CUT_ALIGN = 512


def needed_rows(start: np.ndarray, count: int) -> tuple[int, int]:
    """Panel row span [lo, hi) that a shard with contiguous windows touches (to upload only that); lo is aligned
    down to PANEL_CUT_ALIGN rows."""
    if len(start) == 0:
        return 0, 0
    return (int(start.min()) // PANEL_CUT_ALIGN) * PANEL_CUT_ALIGN, int(start.max()) + int(count)


def slice_window_inputs(inputs: dict, lo: int, hi: int, n_r: int, m: int) -> dict:
    """Shard a `make_kernel_inputs`-style dict (contiguous start/hf_start mode) to windows [lo, hi):
    per-window arrays are cut, panels are cut to the rows the shard reads and offsets rebased."""
    out = dict(inputs)
    start = np.asarray(inputs["start"][lo:hi], dtype=np.int64)
    r0, r1 = needed_rows(start, n_r)
    out["panel"] = inputs["panel"][r0:r1]
    out["start"] = start - r0
    if inputs.get("hf_panel") is not None:
        hs = np.asarray(inputs["hf_start"][lo:hi], dtype=np.int64)
        h0, h1 = needed_rows(hs, m)
        out["hf_panel"] = inputs["hf_panel"][h0:h1]
        out["hf_start"] = hs - h0
        out["w0"] = inputs["w0"][lo:hi]
        out["n0"] = inputs["n0"][lo:hi]
    out["W"] = hi - lo
    return out


def assemble_gathered(parts: list[np.ndarray], ranges: list[tuple[int, int]]) -> np.ndarray:
    """Concatenate per-rank results (possibly padded to a common length) back into window order."""
    return np.concatenate([p[: hi - lo] for p, (lo, hi) in zip(parts, ranges)], axis=0)


# ---------------------------------------------------------------------------------------------------------
# single process, several devices
_PER_WINDOW = ("start", "row_idx", "n_rows", "col_idx", "rf_adj", "hf_start", "hf_row_idx", "hf_count", "w0", "n0",
               "rhs", "shift")


def shard_kwargs(kw: dict, lo: int, hi: int, pad_to: int) -> dict:
    """Windows [lo, hi) of the keyword arguments of `_native.posterior_batch` (any layout): per-window arrays are
    cut, shared panels and row-pair tables are passed through.  The shard is padded to `pad_to` windows by
    repeating its last window (RCCL's gather wants equal counts; `assemble_gathered` drops the padding)."""
    out = {}
    n = hi - lo
    for key, val in kw.items():
        if key in _PER_WINDOW and val is not None:
            a = np.asarray(val)[lo:hi]
            if pad_to > n:
                a = np.concatenate([a, np.repeat(a[-1:], pad_to - n, axis=0)], axis=0)
            out[key] = np.ascontiguousarray(a)
        else:
            out[key] = val
    return out


def run_sharded(group, strategy, k, N, gamma, kw: dict, flags: int = 0, want_aux: bool = False):
    """`_native.posterior_batch` over every device of `group` (a `_native.DeviceGroup`, or anything with
    `.devices`, `.world` and `.gather(batches)` whose devices have `.batch(...)`): contiguous shards, uploads from
    one host thread per device, asynchronous launches, ONE gather of the weights and statuses to device 0.
    Returns (weights [W x k], status [W], aux or None); bit-identical to the unsharded call, because windows
    are independent of each other."""
    kw = dict(kw)
    n_r, m = kw.pop("n_r"), kw.pop("m", 0) or 0
    rhs, shift = kw.pop("rhs", None), kw.pop("shift", None)
    any_per_window = next(v for key, v in kw.items() if key in _PER_WINDOW and v is not None)
    W = len(any_per_window)
    world = group.world
    ranges = partition(W, world)
    w_max = max(hi - lo for lo, hi in ranges)
    if w_max == 0:
        return np.empty((0, k)), np.empty(0, np.int32), (np.empty((0, 8)) if want_aux else None)
    batches = [None] * world
    errors = [None] * world
    if rhs is not None:
        kw["rhs"] = rhs
    if shift is not None:
        kw["shift"] = shift

    def work(r):
        try:
            lo, hi = ranges[r]
            if hi == lo:          # fewer windows than devices: this device repeats the last window of the batch
                lo, hi = W - 1, W
            sub = shard_kwargs(kw, lo, hi, w_max)
            b = group.devices[r].batch(strategy, k, N, n_r, gamma, w_max, m, flags)
            batches[r] = b
            r_rhs, r_shift = sub.pop("rhs", None), sub.pop("shift", None)
            if r_rhs is not None:
                b.set_rhs(r_rhs)
            if r_shift is not None:
                b.set_shift(r_shift)
            b.upload(**sub)
            b.run()               # asynchronous on the device's stream
        except Exception as e:    # noqa: BLE001 - re-raised on the calling thread
            errors[r] = e

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    try:
        for e in errors:
            if e is not None:
                raise e
        wall, sall = group.gather(batches)
        weights = assemble_gathered(list(wall), ranges)
        status = assemble_gathered(list(sall), ranges)
        aux = None
        if want_aux:
            aux = assemble_gathered([b.download(want_aux=True)[2] for b in batches], ranges)
        return weights, status, aux
    finally:
        for b in batches:
            if b is not None:
                b.close()


# ---------------------------------------------------------------------------------------------------------
# one process per device: the control plane
def _send(sock, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n: int) -> bytes:
    chunks = []
    while n:
        c = sock.recv(min(n, 1 << 20))
        if not c:
            raise ConnectionError("control plane: peer closed the connection")
        chunks.append(c)
        n -= len(c)
    return b"".join(chunks)


def _recv(sock) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def _pack_array(a: np.ndarray) -> bytes:
    a = np.ascontiguousarray(a)
    head = json.dumps({"dtype": a.dtype.str, "shape": list(a.shape)}).encode()
    return struct.pack("<I", len(head)) + head + a.tobytes()


def _unpack_array(b: bytes) -> np.ndarray:
    (hl,) = struct.unpack("<I", b[:4])
    head = json.loads(b[4:4 + hl].decode())
    return np.frombuffer(b[4 + hl:], dtype=np.dtype(head["dtype"])).reshape(head["shape"]).copy()


class TcpTransport:
    """Star topology over TCP: rank 0 listens, every other rank holds one connection to it.  Rendezvous: with
    TP_CONTROL_PORT set, rank 0 listens on MASTER_ADDR:TP_CONTROL_PORT (`bench.py` picks a free port when it
    spawns its workers); otherwise - e.g. under `torch.distributed.run`, whose own store owns MASTER_PORT - rank 0
    listens on an ephemeral port and publishes it in a file named after MASTER_ADDR/MASTER_PORT in the temp
    directory (all ranks of this path are on one node)."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world = rank, world
        self.addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.peers: list = []
        self.sock = None
        self._rdzv_file = None
        deadline = time.time() + _TIMEOUT_S
        port_env = os.environ.get("TP_CONTROL_PORT")
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((self.addr, int(port_env) if port_env else 0))
            srv.listen(world)
            srv.settimeout(_TIMEOUT_S)
            if not port_env:
                self._rdzv_file = self._rendezvous_path()
                tmp = self._rdzv_file + f".{os.getpid()}.tmp"
                with open(tmp, "w") as fh:
                    fh.write(str(srv.getsockname()[1]))
                os.replace(tmp, self._rdzv_file)
            conns = {}
            while len(conns) < world - 1:
                c, _ = srv.accept()
                c.settimeout(_TIMEOUT_S)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                hello = _recv(c)
                ok = len(hello) == len(_MAGIC) + 8 and hello[:len(_MAGIC)] == _MAGIC
                r, w = struct.unpack("<II", hello[len(_MAGIC):]) if ok else (0, 0)
                if not ok or w != world or not (1 <= r < world) or r in conns:
                    c.close()              # a stray or stale client: not one of this job's ranks
                    continue
                _send(c, _MAGIC)
                conns[r] = c
            srv.close()
            self.peers = [conns[r] for r in range(1, world)]
        else:
            last = None
            while True:
                if time.time() > deadline:
                    raise TimeoutError(f"control plane: rank {rank} could not reach rank 0 ({last})")
                try:
                    port = int(port_env) if port_env else int(open(self._rendezvous_path()).read().strip())
                    s = socket.create_connection((self.addr, port), timeout=5.0)
                    s.settimeout(_TIMEOUT_S)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send(s, _MAGIC + struct.pack("<II", rank, world))
                    if _recv(s) != _MAGIC:
                        raise ConnectionError("bad handshake")
                    self.sock = s
                    break
                except (OSError, ValueError, ConnectionError) as e:   # rank 0 not up yet, or a stale file
                    last = e
                    time.sleep(0.05)

    def _rendezvous_path(self):
        key = f"{self.addr}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{self.world}"
        return os.path.join(tempfile.gettempdir(), f"tangency_rdzv_{key}")

    # every collective: non-root ranks send one message to rank 0, rank 0 answers when the operation has an answer
    def exchange(self, payload: bytes, reduce_fn, reply: bool):
        """Rank 0 gets `reduce_fn([payload_0, ..., payload_{world-1}])`; with `reply` every rank gets it."""
        if self.rank == 0:
            parts = [payload] + [_recv(c) for c in self.peers]
            out = reduce_fn(parts)
            if reply:
                for c in self.peers:
                    _send(c, out)
            return out
        _send(self.sock, payload)
        return _recv(self.sock) if reply else None

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        if self._rdzv_file:
            try:
                os.unlink(self._rdzv_file)
            except OSError:
                pass
        self.peers, self.sock = [], None


class ControlPlane:
    """Rendezvous / barrier / small host-side collectives of the one-process-per-GPU mode.  world == 1 needs no
    transport at all.  `transport`: an object with `exchange(payload, reduce_fn, reply)` and `close()`."""

    def __init__(self, transport=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self._t = transport
        if self._t is None and self.world > 1:
            self._t = TcpTransport(self.rank, self.world)

    def barrier(self):
        if self._t is not None:
            self._t.exchange(b"", lambda parts: b"", reply=True)

    def _reduce(self, x: float, fn) -> float:
        if self._t is None:
            return float(x)
        out = self._t.exchange(struct.pack("<d", float(x)),
                               lambda parts: struct.pack("<d", fn(struct.unpack("<d", p)[0] for p in parts)), reply=True)
        return struct.unpack("<d", out)[0]

    def max(self, x: float) -> float:
        return self._reduce(x, max)

    def sum(self, x: float) -> float:
        return self._reduce(x, lambda it: float(np.sum(np.fromiter(it, dtype=np.float64))))

    def bcast_bytes(self, payload: bytes | None, nbytes: int, src: int = 0) -> bytes:
        if self._t is None:
            return payload
        if src != 0:
            raise ValueError("the control plane broadcasts from rank 0")
        out = self._t.exchange(payload if self.rank == 0 else b"", lambda parts: parts[0], reply=True)
        if len(out) != nbytes:
            raise ValueError(f"broadcast of {len(out)} bytes, {nbytes} expected")
        return bytes(out)

    def gather_host(self, arr: np.ndarray, root: int = 0):
        """Host gather of equal-shaped arrays to rank 0 (checksums, rehearsals and CPU tests)."""
        if self._t is None:
            return [np.asarray(arr)]
        if root != 0:
            raise ValueError("the control plane gathers to rank 0")
        got = []
        self._t.exchange(_pack_array(np.asarray(arr)), lambda parts: got.extend(parts) or b"", reply=False)
        return [_unpack_array(p) for p in got] if self.rank == 0 else None

    def close(self):
        if self._t is not None:
            self._t.close()
            self._t = None


def init_rccl(dev, cp: ControlPlane):
    """Create the RCCL communicator of `dev` (a `_native.Device`): rank 0 draws the id, the control
    plane broadcasts it, every rank calls ncclCommInitRank.  Raises on failure: there is no other transport."""
    from . import _native
    uid = _native.Device.comm_unique_id() if cp.rank == 0 else None
    uid = cp.bcast_bytes(uid, _native.UNIQUE_ID_BYTES, src=0)
    dev.comm_init(uid, cp.rank, cp.world)
