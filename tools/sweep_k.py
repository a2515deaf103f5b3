"""Throughput of the posterior kernels across universe sizes (N=250 daily rows, 77 intraday returns; k > 249 uses
5 intraday days so that S1 has full rank).  GPU box: python tools/sweep_k.py [k ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from incorporating_different_sources_amd import _native, synthetic

ks = [int(a) for a in sys.argv[1:]] or [8, 15, 16, 31, 47, 63, 79, 95, 100, 111, 127, 143, 159, 175, 191, 207, 223, 239, 240, 320, 500]
dev = _native.default_device()
for k in ks:
    N, hf_days = 250, (1 if k < 240 else 5)
    W = max(256, min(10000, int(4e6 / (k * k)) * 16))
    inp = synthetic.make_kernel_inputs(k, N, W, seed=1000 + k, hf_days=hf_days)
    b = dev.batch("conjugate", k, N, inp["n_r"], 5.0, W, inp["m"])
    b.upload(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
    for _ in range(2): b.run()
    dev.synchronize(); dev.region_begin()
    reps = 5
    for _ in range(reps): b.run()
    ms = dev.region_end() / reps
    w, s, _ = b.download(want_aux=False)
    flops = (inp["n_r"] + inp["m"]) * k * (k + 1) + k ** 3 / 3 + 6 * k * k
    li = dev.last_launch()
    print(f"k={k:4d} W={W:6d} {ms:8.3f} ms {W / ms * 1e3:12.0f} win/s  {flops * W / ms / 1e9:6.2f} TF  frac {flops * W / ms / 1e9 / 78.6:5.3f}  bad={int((s != 0).sum())} launch={li}", flush=True)
    b.close()
