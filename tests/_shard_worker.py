"""Worker of tests/test_shard_gloo.py: one rank of a world_size-2 gloo job on CPU.  The compute
stand-in is the oracle (tests may call it); what is under test is the package's sharding, control
plane and gather assembly."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from incorporating_different_sources_amd import shard, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    out_path = sys.argv[1]
    W, k, N = 37, 12, 30
    cp = shard.ControlPlane("gloo")
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
    cp.barrier()
    if cp.rank == 0:
        full = shard.assemble_gathered(parts, ranges)
        ref, _, _ = oracle.posterior_batch(
            "conjugate", k, N, 5.0, panel=inp["panel"], start=inp["start"], n_r=inp["n_r"],
            hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], m=inp["m"], w0=inp["w0"], n0=inp["n0"])
        json.dump({"max_abs_diff": float(np.abs(full - ref).max()), "shape": list(full.shape),
                   "uid_ok": uid == bytes(range(128)), "tmax": tmax, "world": cp.world}, open(out_path, "w"))
    else:
        assert uid == bytes(range(128))
    cp.close()


if __name__ == "__main__":
    main()
