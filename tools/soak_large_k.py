"""Randomised soak of the large-k path (contiguous layout, every sharing scheme on): universe sizes 240..1100, both strategies,
1..12 intraday days, 6..70 windows, arenas small enough to cut a run into several sub-batches - against the C oracle on a
sample of windows (flat 1e-10 on the weights, well-posed shapes only) and against the same batch without shared sums.
GPU box: python tools/soak_large_k.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from incorporating_different_sources_amd import _native as native, synthetic
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
dev = native.default_device()
worst = 0.0
t0 = time.time()
for c in range(cases):
    k = int(rng.choice([rng.integers(240, 330), rng.integers(330, 700), rng.integers(700, 1100)], p=[0.5, 0.35, 0.15]))
    strat = str(rng.choice(["conjugate", "jeffreys"], p=[0.7, 0.3]))
    hf_days = int(rng.integers(1, 13))
    N = int(1.6 * k + rng.integers(20, 200)) if strat == "jeffreys" else int(max(30, k - 78 * hf_days + rng.integers(40, 300)))
    W = int(rng.integers(6, 70))
    inp = synthetic.make_kernel_inputs(k, N, W, seed=int(rng.integers(1 << 30)), hf_days=hf_days)
    kw = dict(panel=inp["panel"], start=inp["start"], n_r=inp["n_r"])
    if strat == "conjugate":
        kw.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], m=inp["m"], w0=inp["w0"], n0=inp["n0"])
    arena = int(rng.choice([0, 0, 64, 256]))
    minblk = int(rng.choice([2, 6]))
    dev.set_option("tiled_arena_mib", arena)
    dev.set_option("hf_share_min_blocks", minblk)
    try:
        got, st, aux = native.posterior_batch(strat, k, N, 5.0, **kw)
    finally:
        dev.set_option("tiled_arena_mib", 0)
        dev.set_option("hf_share_min_blocks", 6)
    plain, pst, _ = native.posterior_batch(strat, k, N, 5.0, flags=native.FLAG_NO_SHARED_GRAM, **kw)
    sel = np.unique(rng.integers(0, W, 3))
    sub = {key: (val[sel] if key in ("start", "hf_start", "w0", "n0") else val) for key, val in kw.items()}
    ref, rstat, _ = oracle.posterior_batch_c(strat, k, N, 5.0, **sub)
    ok = (rstat == 0)
    assert (st[sel] == rstat).all() and (pst == st).all(), (c, k, N, strat, st, rstat)
    d_or = float(np.abs(got[sel][ok] - ref[ok]).max()) if ok.any() else 0.0
    d_pl = float(np.abs(got[st == 0] - plain[st == 0]).max()) if (st == 0).any() else 0.0
    worst = max(worst, d_or)
    flag = "" if d_or <= 1e-10 else "   <-- above 1e-10"
    print(f"{c:3d} k={k:4d} N={N:4d} {strat:9s} hf_days={hf_days:2d} W={W:2d} arena={arena:3d} minblk={minblk} ok={int(ok.sum())}/{len(sel)} "
          f"|hip-oracle| {d_or:.1e}  |shared-plain| {d_pl:.1e}  max|w| {np.abs(ref).max():.2g}{flag}", flush=True)
print(f"worst |hip - oracle| {worst:.2e} over {cases} cases in {time.time() - t0:.0f} s")
