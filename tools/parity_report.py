"""Observed max |hip - oracle| per shape (GPU box): the numbers DESIGN.md quotes next to the test tolerances.
Usage: python tools/parity_report.py > gpurun_out/parity_report.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incorporating_different_sources_amd import _native, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

fused = [(1, 8, 1), (2, 9, 1), (7, 20, 1), (15, 40, 1), (16, 40, 1), (17, 60, 1), (31, 70, 1), (32, 70, 1), (47, 100, 1),
         (48, 120, 1), (64, 150, 1), (95, 200, 1), (96, 250, 1), (100, 250, 1), (111, 250, 1), (112, 250, 2), (128, 300, 2),
         (150, 320, 2), (191, 400, 3), (192, 400, 3), (200, 420, 3), (224, 460, 3), (239, 500, 4)]
tiled = [(240, 300, 2, "conjugate"), (255, 300, 2, "conjugate"), (256, 300, 2, "conjugate"), (300, 700, 1, "jeffreys"),
         (319, 250, 3, "conjugate"), (320, 250, 3, "conjugate"), (500, 250, 5, "conjugate"), (511, 260, 5, "conjugate"),
         (512, 1100, 1, "jeffreys"), (640, 400, 6, "conjugate"), (1000, 500, 22, "conjugate"), (2047, 300, 24, "conjugate")]


def one(k, N, hf_days, strat, W, seed):
    inp = synthetic.make_kernel_inputs(k, N, W, seed=seed, hf_days=hf_days)
    kw = dict(panel=inp["panel"], start=inp["start"], n_r=inp["n_r"])
    if strat == "conjugate":
        kw.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], m=inp["m"], w0=inp["w0"], n0=inp["n0"])
    ref, rstat, raux = oracle.posterior_batch_c(strat, k, N, 5.0, **kw)
    wts, status, aux = _native.posterior_batch(strat, k, N, 5.0, **kw)
    err = np.abs(wts - ref).max()
    scale = np.abs(ref).max()
    aerr = (np.abs(aux[:, :6] - raux[:, :6]) / np.maximum(np.abs(raux[:, :6]), 1e-300)).max() if strat == "conjugate" else 0.0
    print(f"k={k:5d} N={N:5d} {strat:9s} max|w| {scale:10.3e}  max|hip-oracle| {err:9.2e}  rel-to-max {err / scale:9.2e}  "
          f"aux rel {aerr:9.2e}  status {'ok' if (status == rstat).all() else 'DIFF'}", flush=True)


for k, N, hf in fused:
    for strat in ("conjugate", "jeffreys"):
        one(k, N, hf, strat, 5, 777000 + k)
for k, N, hf, strat in tiled:
    one(k, N, hf, strat, 3, 880000 + k)
