"""Kernel throughput in the index modes real backtests use (row_idx + col_idx + rf_adj, intraday row_idx),
next to the contiguous mode of the synthetic benchmark.  GPU box: python tools/time_index_modes.py [lib.so ...]"""
import os, subprocess, sys

CODE = r'''
import numpy as np, time
from incorporating_different_sources_amd import _native, synthetic
k, N, W = 100, 250, 10000
inp = synthetic.make_kernel_inputs(k, N, W, seed=20240002)
n_r, m = inp["n_r"], inp["m"]
dev = _native.default_device()
def run(kw, mm):
    b = dev.batch("conjugate", k, N, n_r, 5.0, W, mm)
    b.upload(**kw)
    for _ in range(3): b.run()
    dev.synchronize(); dev.region_begin()
    for _ in range(20): b.run()
    ms = dev.region_end() / 20
    w, s, _ = b.download(want_aux=False); b.close()
    return ms, w
base = dict(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
ms0, w0 = run(base, m)
# the same windows through explicit row / column indices (identity gather) and a zero risk-free adjustment
rows = (inp["start"][:, None] + np.arange(n_r)[None, :]).astype(np.int32)
hrows = (inp["hf_start"][:, None] + np.arange(m)[None, :]).astype(np.int32)
cols = np.tile(np.arange(k, dtype=np.int32), (W, 1))
idx = dict(panel=inp["panel"], row_idx=rows, col_idx=cols, rf_adj=np.zeros((W, n_r)), hf_panel=inp["hf_panel"],
           hf_row_idx=hrows, w0=inp["w0"], n0=inp["n0"])
ms1, w1 = run(idx, m)
print(f"contiguous {ms0:.3f} ms ({W/ms0*1e3:.0f} win/s)   row_idx+col_idx+rf_adj {ms1:.3f} ms ({W/ms1*1e3:.0f} win/s)   max|diff| {np.abs(w0-w1).max():.2e}")
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["libtangency.so"]
for lib in libs:
    env = dict(os.environ, TANGENCY_LIB=os.path.join(root, "incorporating_different_sources_amd", lib), PYTHONPATH=root)
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", CODE], env=env, check=False)
