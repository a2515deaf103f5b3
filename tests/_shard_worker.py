"""Worker of tests/test_shard_gloo.py: one rank of a world_size-2 job on CPU.  The compute stand-in is the
oracle (tests may call it); what is under test is the package's sharding, control plane and gather assembly.
argv[2] selects the control-plane transport: "tcp" (the package's own, standard library sockets) or "gloo"
(torch.distributed injected through `ControlPlane(transport=...)` - torch stays out of the package)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from incorporating_different_sources_amd import shard, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402


class GlooTransport:
    """`exchange` over torch.distributed (gloo): gather the payloads to rank 0, broadcast the reduced answer."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange(self, payload, reduce_fn, reply):
        torch, dist = self.torch, self.dist
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(sizes, torch.tensor([len(payload)], dtype=torch.int64))
        n = max(int(s.item()) for s in sizes)
        buf = torch.zeros(max(n, 1), dtype=torch.uint8)
        if payload:
            buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        if self.rank == 0:
            outs = [torch.zeros_like(buf) for _ in range(self.world)]
            dist.gather(buf, gather_list=outs, dst=0)
            out = reduce_fn([bytes(o.numpy().tobytes()[: int(s.item())]) for o, s in zip(outs, sizes)])
        else:
            dist.gather(buf, gather_list=None, dst=0)
            out = None
        if not reply:
            return out
        ln = torch.tensor([len(out) if self.rank == 0 else 0], dtype=torch.int64)
        dist.broadcast(ln, src=0)
        t = torch.zeros(max(int(ln.item()), 1), dtype=torch.uint8)
        if self.rank == 0 and out:
            t[: len(out)] = torch.frombuffer(bytearray(out), dtype=torch.uint8)
        dist.broadcast(t, src=0)
        return bytes(t.numpy().tobytes()[: int(ln.item())])

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()


def main():
    out_path, transport = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "tcp")
    W, k, N = 37, 12, 30
    cp = shard.ControlPlane(transport=GlooTransport() if transport == "gloo" else None)
    inp = synthetic.make_kernel_inputs(k, N, W, seed=4242)
    ranges = shard.partition(W, cp.world)
    lo, hi = ranges[cp.rank]
    mine = shard.slice_window_inputs(inp, lo, hi, inp["n_r"], inp["m"])
    wts, status, _ = oracle.posterior_batch(
        "conjugate", k, N, 5.0, panel=mine["panel"], start=mine["start"], n_r=inp["n_r"],
        hf_panel=mine["hf_panel"], hf_start=mine["hf_start"], m=inp["m"], w0=mine["w0"], n0=mine["n0"])
    # equal-shaped gather: pad to the largest shard
    wmax = max(h - l for l, h in ranges)
    pad = np.zeros((wmax, k)); pad[: hi - lo] = wts
    uid = cp.bcast_bytes(bytes(range(128)) if cp.rank == 0 else None, 128, src=0)
    parts = cp.gather_host(pad, root=0)
    tmax = cp.max(float(cp.rank + 1))
    tsum = cp.sum(float(cp.rank + 1))
    cp.barrier()
    if cp.rank == 0:
        full = shard.assemble_gathered(parts, ranges)
        ref, _, _ = oracle.posterior_batch(
            "conjugate", k, N, 5.0, panel=inp["panel"], start=inp["start"], n_r=inp["n_r"],
            hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], m=inp["m"], w0=inp["w0"], n0=inp["n0"])
        json.dump({"max_abs_diff": float(np.abs(full - ref).max()), "shape": list(full.shape),
                   "uid_ok": uid == bytes(range(128)), "tmax": tmax, "tsum": tsum, "world": cp.world,
                   "transport": transport}, open(out_path, "w"))
    else:
        assert uid == bytes(range(128))
    cp.close()


if __name__ == "__main__":
    main()
