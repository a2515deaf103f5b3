"""Throughput of the register-tile kernels across universe sizes (N=250 daily rows, 77 intraday returns; k > 249 uses
5 intraday days so that S1 has full rank), every kernel built for a size in ONE run on one box: the multi-wave kernel
(option wave_kernel = 0), one wave per window (1), two / four waves per window (2); with the shared Gram sums of the
contiguous layout and without them (TP_FLAG_NO_SHARED_GRAM: what every row costs when it goes through the MFMAs).
GPU box: python tools/sweep_k.py [--windows W] [k ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from incorporating_different_sources_amd import _native, synthetic

args = sys.argv[1:]
Wfix = 0
if args and args[0] == "--windows":
    Wfix = int(args[1]); args = args[2:]
ks = [int(a) for a in args] or [8, 15, 16, 31, 47, 63, 79, 95, 100, 111, 127, 143, 150, 159, 175, 191, 192, 207, 223, 239, 240, 320, 500]
dev = _native.default_device()
NAMES = {0: "multi-wave", 1: "one-wave", 2: "two/four-wave"}
for k in ks:
    N, hf_days = 250, (1 if k < 240 else 5)
    nt = (k + 1 + 15) // 16
    W = Wfix or max(256, min(10000, int(4e6 / (k * k)) * 16))
    inp = synthetic.make_kernel_inputs(k, N, W, seed=1000 + k, hf_days=hf_days)
    flops = (inp["n_r"] + inp["m"]) * k * (k + 1) + k ** 3 / 3 + 6 * k * k
    choices = [-1] if k > 239 else [c for c in (0, 1, 2) if (c == 0 or (c == 1 and nt <= 9) or (c == 2 and nt >= 7))]
    for flags, tag in ((0, "shared"), (_native.FLAG_NO_SHARED_GRAM, "no-sharing")):
        ref = None
        for c in choices:
            dev.set_option("wave_kernel", c)
            b = dev.batch("conjugate", k, N, inp["n_r"], 5.0, W, inp["m"], flags=flags)
            b.upload(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
            for _ in range(2): b.run()
            dev.synchronize(); dev.region_begin()
            reps = 5
            for _ in range(reps): b.run()
            ms = dev.region_end() / reps
            w, s, _ = b.download(want_aux=False)
            li = dev.last_launch()
            diff = 0.0 if ref is None else float(np.abs(w - ref).max())
            ref = w if ref is None else ref
            print(f"k={k:4d} W={W:6d} {tag:10s} {NAMES.get(c, 'tiled'):14s} {ms:8.3f} ms {W / ms * 1e3:12.0f} win/s  {flops * W / ms / 1e9:6.2f} TF  "
                  f"frac {flops * W / ms / 1e9 / 78.6:5.3f}  bad={int((s != 0).sum())} block={li['block']} lds={li['lds_bytes']} d={diff:.1e}", flush=True)
            b.close()
dev.set_option("wave_kernel", -1)
