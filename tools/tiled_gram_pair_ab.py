import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from incorporating_different_sources_amd import _native, synthetic
dev = _native.default_device()
for cfg, W in ((3, 4096), (5, 384)):
    shp = synthetic.config_shapes(cfg)
    k, N, n_r, m = shp["k"], shp["N"], shp["n_r"], shp["m"]
    inp = synthetic.make_kernel_inputs(k, N, W, seed=shp["seed"], hf_days=shp["hf_days"])
    flops = (n_r + m) * k * (k + 1) + k ** 3 / 3 + 6 * k * k
    ref = None
    for flags, tag in ((0, "shared"), (4, "no-sharing")):
        for tw in (-1, 2, -1, 2):
            dev.set_option("tiled_wave", tw)
            b = dev.batch("conjugate", k, N, n_r, 5.0, W, m, flags=flags)
            b.upload(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
            b.run(); dev.synchronize(); dev.region_begin()
            for _ in range(3): b.run()
            ms = dev.region_end() / 3
            w, s, _ = b.download(want_aux=False)
            same = ref is None or bool(np.array_equal(w, ref)); ref = w if ref is None else ref
            print(f"k={k} W={W} {tag:10s} tiled_wave={tw:2d}: {ms:8.3f} ms frac {flops*W/ms/1e9/78.6:5.3f} bad={int((s!=0).sum())} identical_to_first={same} maxdiff={float(np.abs(w-ref).max()):.1e}", flush=True)
            b.close()
        ref = None
dev.set_option("tiled_wave", -1)
