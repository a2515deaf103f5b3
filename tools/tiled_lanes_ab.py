"""Large-k path: depth-first sub-batches (options tiled_lanes / tiled_arena_mib) against the default one-lane 32 GiB arena,
same inputs, one process.  GPU box: python tools/tiled_lanes_ab.py [config id] [windows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from incorporating_different_sources_amd import _native, synthetic

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
shp = synthetic.config_shapes(cfg)
k, N, n_r, m = shp["k"], shp["N"], shp["n_r"], shp["m"]
inp = synthetic.make_kernel_inputs(k, N, W, seed=shp["seed"], hf_days=shp["hf_days"])
flops = (n_r + m) * k * (k + 1) + k ** 3 / 3 + 6 * k * k
dev = _native.default_device()
ref = None
combos = [(0, 0)] + [(l, mib) for mib in (32, 64, 128, 256, 1024) for l in (1, 2, 3, 4)]
for lanes, mib in combos:
    dev.set_option("tiled_lanes", lanes).set_option("tiled_arena_mib", mib)
    b = dev.batch("conjugate", k, N, n_r, 5.0, W, m)
    b.upload(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
    b.run(); dev.synchronize()
    dev.region_begin()
    reps = 3
    for _ in range(reps):
        b.run()
    ms = dev.region_end() / reps
    w, s, _ = b.download(want_aux=False)
    same = True if ref is None else bool(np.array_equal(w, ref))
    ref = w if ref is None else ref
    print(f"k={k} W={W} lanes={lanes} arena_mib={mib or 'default'}: {ms:8.3f} ms  {W / ms * 1e3:10.0f} win/s  frac {flops * W / ms / 1e9 / 78.6:5.3f} "
          f"in-flight/lane={dev.last_launch()['grid']} bad={int((s != 0).sum())} identical={same}", flush=True)
    b.close()
dev.set_option("tiled_lanes", 0).set_option("tiled_arena_mib", 0)
