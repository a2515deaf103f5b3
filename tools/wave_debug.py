"""Diagnostic: the one-wave-per-window kernel against the multi-wave kernel, matrix by matrix (GPU box).
usage: python tools/wave_debug.py [k N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incorporating_different_sources_amd import _native, synthetic

k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 250
W = 40
inp = synthetic.make_kernel_inputs(k, N, W, seed=4242)
dev = _native.default_device()


def run(wave, strat="conjugate"):
    os.environ["TP_WAVE_KERNEL"] = "1" if wave else "0"
    b = _native.Batch(dev, strat, k, N, inp["n_r"], 5.0, W, inp["m"] if strat == "conjugate" else 0)
    kw = dict(panel=inp["panel"], start=inp["start"])
    if strat == "conjugate":
        kw.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
    b.upload(**kw)
    out = {}
    for what in ("prior", "gram", "posterior"):
        if strat != "conjugate" and what == "prior":
            continue
        try:
            out[what] = b.download_matrix(3, what)
        except Exception as e:   # noqa
            out[what] = repr(e)
    b.run()
    w, s, aux = b.download()
    out["weights"], out["status"], out["aux"] = w, s, aux
    b.close()
    return out


for strat in ("conjugate", "jeffreys"):
    a, b = run(False, strat), run(True, strat)
    print("==", strat, "k", k, "N", N)
    for key in a:
        x, y = a[key], b[key]
        if isinstance(x, tuple):
            for i, (xi, yi) in enumerate(zip(x, y)):
                print(f"  {key}[{i}] max|diff| {np.abs(np.asarray(xi) - np.asarray(yi)).max():.3e}  scale {np.abs(np.asarray(xi)).max():.3e}")
        elif isinstance(x, np.ndarray):
            d = np.abs(x.astype(float) - y.astype(float))
            print(f"  {key}: max|diff| {np.nanmax(d):.3e} scale {np.abs(x).max():.3e} nan {np.isnan(y.astype(float)).sum()}")
            if key == "aux":
                print("   old", x[3], "\n   new", y[3])
        else:
            print("  ", key, x, y)
