"""Diagnostic: per-phase share of the fused kernel's per-window time, from in-kernel s_memtime
stamps.  Needs a TP_STAMP build:  make -C incorporating_different_sources_amd/csrc clean && \
make -j8 -C incorporating_different_sources_amd/csrc TP_STAMP=1   (rebuild without it afterwards)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incorporating_different_sources_amd import _native, synthetic

# usage: phase_breakdown.py <config id | k=K> [windows] [flags]     (k=K: N=250, one intraday day; flags 4 = no shared Gram)
arg = sys.argv[1] if len(sys.argv) > 1 else "2"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if arg.startswith("k="):
    kk = int(arg[2:])
    shp = dict(k=kk, N=250, n_r=249, m=77, seed=777 + kk, hf_days=1)
else:
    shp = synthetic.config_shapes(int(arg))
inp = synthetic.make_kernel_inputs(shp["k"], shp["N"], W, shp["seed"], hf_days=shp["hf_days"])
dev = _native.Device(0)
b = dev.batch("conjugate", shp["k"], shp["N"], shp["n_r"], 5.0, W, shp["m"], flags=flags)
b.upload(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
for _ in range(3):
    b.run()
dev.synchronize()
st = b.debug_stamps()
print("kernel_ms", dev.last_timing()["kernel_ms"], dev.last_launch())
loop = st[:, 8:24].reshape(-1, 4, 4).astype(np.float64)
fseg = st[:, 24:32].reshape(-1, 2, 4).astype(np.float64)
st = st[:, :8]
d = np.diff(st, axis=1).astype(np.float64)
names = ["A means", "B hf gram", "C scale", "D daily gram", "F cholesky", "G backsolve", "H output"]
tot = (st[:, 7] - st[:, 0]).astype(np.float64)
print(f"per-window wall (s_memtime ticks @100MHz): median {np.median(tot):.0f} ticks = {np.median(tot)*10:.0f} ns")
for i, n in enumerate(names):
    print(f"  {n:14s} median {np.median(d[:, i]):8.0f} ticks  share {np.median(d[:, i]) / np.median(tot) * 100:5.1f}%")
span = st[:, 7].max() - st[:, 0].min()
print("launch span ticks", span, "=> ms", span / 1e5)

print("daily Gram loop, summed cycles per window by segment (median over windows), per wave:")
for wv in range(4):
    m = np.median(loop[:, wv, :], axis=0)
    print(f"  wave {wv}: issue-loads {m[0]:8.0f}  mfma {m[1]:8.0f}  finish+lds-write {m[2]:8.0f}  barrier {m[3]:8.0f}  total {m.sum():8.0f}")

print("factorisation, summed cycles per window by block-step segment (median), waves 0 and 1:")
for wv in range(2):
    m = np.median(fseg[:, wv, :], axis=0)
    print(f"  wave {wv}: hand-over+barrier {m[0]:8.0f}  elimination+barrier {m[1]:8.0f}  trsm+barrier {m[2]:8.0f}  trailing {m[3]:8.0f}  total {m.sum():8.0f}")

# one-wave-per-window kernel: extra stamps inside phase D (slots 32..34: before the edge rows, after them, after the table read)
full = b.debug_stamps()
if full[:, 32].any():
    x = full[:, [3, 32, 33, 34, 4]].astype(np.float64)
    dd = np.diff(x, axis=1)
    print("wave kernel, phase D split (median ticks): setup %.0f  edge rows %.0f  table slot %.0f  rest %.0f" % tuple(np.median(dd, axis=0)))
