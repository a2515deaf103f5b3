"""Accuracy of the shared intraday sums of the large-k path at the far end of the universe sizes, against the C oracle and
beside the two-pass form (TP_FLAG_NO_SHARED_GRAM).  GPU box: python tools/hf_share_accuracy.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incorporating_different_sources_amd import _native as native, synthetic
from oracle import oracle
for (k, N, hf_days, W) in [(2047, 300, 24, 6), (2047, 2200, 8, 6), (1500, 400, 24, 6)]:
    inp = synthetic.make_kernel_inputs(k, N, W, seed=880000 + k, hf_days=hf_days)
    kw = dict(panel=inp["panel"], start=inp["start"], n_r=inp["n_r"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], m=inp["m"], w0=inp["w0"], n0=inp["n0"])
    wts, st, aux = native.posterior_batch("conjugate", k, N, 5.0, **kw)
    pl, pst, _ = native.posterior_batch("conjugate", k, N, 5.0, flags=native.FLAG_NO_SHARED_GRAM, **kw)
    sel = np.array([0, W - 1])
    sub = {key: (val[sel] if key in ("start", "hf_start", "w0", "n0") else val) for key, val in kw.items()}
    ref, rstat, _ = oracle.posterior_batch_c("conjugate", k, N, 5.0, **sub)
    print(k, N, hf_days, "status", st.tolist(), "shared-vs-oracle", np.abs(wts[sel] - ref).max(), "two-pass-vs-oracle", np.abs(pl[sel] - ref).max(), "max|w|", np.abs(ref).max(), "shared==plain", np.array_equal(wts, pl), flush=True)
