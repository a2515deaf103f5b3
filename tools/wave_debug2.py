"""Diagnostic (GPU box): which windows of a batch differ between the two register-tile kernels.  usage: wave_debug2.py k N W"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incorporating_different_sources_amd import _native, synthetic

k, N, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
inp = synthetic.make_kernel_inputs(k, N, W, seed=1000 + k)
dev = _native.default_device()


def run(wave):
    os.environ["TP_WAVE_KERNEL"] = "1" if wave else "0"
    b = _native.Batch(dev, "conjugate", k, N, inp["n_r"], 5.0, W, inp["m"])
    b.upload(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
    b.run()
    out = b.download()
    b.close()
    return out


w0, s0, a0 = run(False)
for rep in range(3):
    w1, s1, a1 = run(True)
    d = np.abs(w1 - w0).max(axis=1)
    badw = np.nonzero((d > 1e-9) | (s1 != s0) | ~np.isfinite(d))[0]
    print(f"rep {rep}: status hist {np.bincount(s1, minlength=4)} windows off {len(badw)} first {badw[:12]}  max diff {np.nanmax(d):.3e}")
    if len(badw):
        i = badw[0]
        print("   aux old", a0[i][:6], "\n   aux new", a1[i][:6])
        print("   window mod 8:", np.bincount(badw % 8, minlength=8), " per-xcd block:", np.bincount((badw * 8) // W, minlength=8))
