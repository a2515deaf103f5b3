"""Multi-GPU host logic: rebalancing windows shard embarrassingly across the GPUs of one node.

One process per GPU.  Windows are split into contiguous ranges (rank r gets `partition(W, world)[r]`),
each rank runs its range through its own `_native.Device`, and ONE gather of the `[W_local x k]`
weights goes to rank 0 over RCCL/xGMI (`tp_batch_gather`).  There is no other data-path collective.

The control plane (rendezvous, barrier, the 128-byte RCCL id, timing reductions) rides on
`torch.distributed` with the `gloo` backend when the process was started by `torch.distributed.run`;
torch is plumbing here, nothing is computed with it.
"""
from __future__ import annotations

import os

import numpy as np


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


def needed_rows(start: np.ndarray, count: int) -> tuple[int, int]:
    """Panel row span [lo, hi) that a shard with contiguous windows touches (to upload only that)."""
    if len(start) == 0:
        return 0, 0
    return int(start.min()), int(start.max()) + int(count)


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


class ControlPlane:
    """Rendezvous / barrier / small host-side collectives.  world == 1 needs no torch at all."""

    def __init__(self, backend: str = "gloo"):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, x: float) -> float:
        if self._dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, x: float) -> float:
        if self._dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def bcast_bytes(self, payload: bytes | None, nbytes: int, src: int = 0) -> bytes:
        if self._dist is None:
            return payload
        import torch
        t = torch.zeros(nbytes, dtype=torch.uint8)
        if self.rank == src:
            t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
        self._dist.broadcast(t, src=src)
        return bytes(t.numpy().tobytes())

    def gather_host(self, arr: np.ndarray, root: int = 0):
        """Host-staged gather of equal-shaped arrays (fallback transport and CPU tests)."""
        if self._dist is None:
            return [arr]
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.rank == root:
            outs = [torch.empty_like(t) for _ in range(self.world)]
            self._dist.gather(t, gather_list=outs, dst=root)
            return [o.numpy() for o in outs]
        self._dist.gather(t, gather_list=None, dst=root)
        return None

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def init_rccl(dev, cp: ControlPlane):
    """Create the RCCL communicator of `dev` (a `_native.Device`): rank 0 draws the id, the control
    plane broadcasts it, every rank calls ncclCommInitRank."""
    from . import _native
    uid = _native.Device.comm_unique_id() if cp.rank == 0 else None
    uid = cp.bcast_bytes(uid, _native.UNIQUE_ID_BYTES, src=0)
    dev.comm_init(uid, cp.rank, cp.world)
