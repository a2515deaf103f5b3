"""Child process of tests/test_gpu_lifetime.py: leaves device objects alive in the ways a failing backtest does and
then ends.  The parent checks the exit code: the Python exception's (1) or sys.exit's, never a signal.

    python tests/_exit_worker.py raise | exit3 | cycle | global | pinned
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from incorporating_different_sources_amd import _native, synthetic  # noqa: E402

KEEP = []


def make(with_comm=True, asynchronous=False):
    inp = synthetic.make_kernel_inputs(20, 50, 64, seed=7)
    dev = _native.Device(0)
    if with_comm:
        dev.comm_init(_native.Device.comm_unique_id(), 0, 1)          # a one-rank RCCL communicator, never destroyed
    b = dev.batch("conjugate", 20, 50, inp["n_r"], 5.0, 64, inp["m"])
    kw = dict(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"],
              n0=inp["n0"])
    if asynchronous:
        pinned = {k: _native.pinned_copy(v) for k, v in kw.items()}
        b.upload_async(**pinned)
    else:
        b.upload(**kw)
    b.run()
    if with_comm:
        b.gather_async(root=0)                                          # requested, not yet on its stream
        b.run()
    w, s, _ = b.download()
    assert np.isfinite(w).all() and (s == 0).all()
    return dev, b


def main(mode):
    if mode == "raise":
        dev, b = make()                     # locals of a frame the traceback keeps alive
        raise RuntimeError("boom: a backtest failed with device objects alive")
    if mode == "exit3":
        dev, b = make()
        sys.exit(3)
    if mode == "cycle":
        dev, b = make()
        cyc = {"dev": dev, "b": b}
        cyc["self"] = cyc                   # a reference cycle: only the cyclic collector (or nobody) finalises it
        KEEP.append(cyc)
        raise RuntimeError("boom")
    if mode == "global":
        _native.default_device()            # the module-global handle plus a leaked batch on it
        inp = synthetic.make_kernel_inputs(10, 40, 8, seed=1)
        b = _native.default_device().batch("jeffreys", 10, 40, inp["n_r"], 5.0, 8, 0)
        b.upload(panel=inp["panel"], start=inp["start"])
        b.run()
        KEEP.append(b)
        raise RuntimeError("boom")
    if mode == "pinned":
        dev, b = make(with_comm=False, asynchronous=True)
        KEEP.append((dev, b))
        raise RuntimeError("boom")
    if mode == "clean":
        dev, b = make()
        b.close()
        dev.close()
        return
    raise SystemExit(f"unknown mode {mode}")


if __name__ == "__main__":
    main(sys.argv[1])
