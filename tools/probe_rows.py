import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from incorporating_different_sources_amd import _native, synthetic
dev = _native.default_device()
for N in (250, 33, 17):
    k, W = 100, 10000
    inp = synthetic.make_kernel_inputs(k, N, W, seed=1000 + k)
    for strat in ("conjugate", "jeffreys"):
        b = dev.batch(strat, k, N, inp["n_r"], 5.0, W, inp["m"] if strat == "conjugate" else 0)
        kw = dict(panel=inp["panel"], start=inp["start"])
        if strat == "conjugate":
            kw.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
        b.upload(**kw)
        for _ in range(2): b.run()
        dev.synchronize(); dev.region_begin()
        for _ in range(5): b.run()
        ms = dev.region_end() / 5
        print(f"N={N} {strat}: {ms:.3f} ms per {W} windows", flush=True)
        b.close()
