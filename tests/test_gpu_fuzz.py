"""Seeded random shapes and layouts through the C-ABI against the oracle (-m gpu): universe sizes over every kernel family
(one wave, two / four waves with the data-flow factorisation, tiled), window lengths, intraday lengths, ragged row counts,
contiguous / index layouts with gathered columns and per-row risk-free adjustments, both strategies, batches small enough
for the C oracle.  The flat 1e-10 of the north star on the weights; statuses equal; conjugate aux to 1e-11."""
import numpy as np
import pytest

from oracle import oracle
from incorporating_different_sources_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from incorporating_different_sources_amd import _native
    return _native


def _case(rng):
    family = rng.choice(["one", "two", "four", "tiled"], p=[0.35, 0.3, 0.25, 0.1])
    k = int({"one": rng.integers(1, 144), "two": rng.integers(144, 192), "four": rng.integers(192, 240),
             "tiled": rng.integers(240, 330)}[family])
    strat = str(rng.choice(["conjugate", "jeffreys"]))
    hf_days = int(rng.integers(1, 4)) if k < 200 else int(rng.integers(3, 6))
    # Jeffreys: rows well above the universe size (the flat 1e-10 presumes a well-posed J: at a rank margin of 14 - k = 195
    # over 209 rows - LU (oracle, reference) and Cholesky differ by 2e-10 on weights of order 100, 8e-12 relative)
    N = int(1.7 * k + rng.integers(10, 60)) if strat == "jeffreys" else int(max(8, k - 78 * hf_days + rng.integers(20, 120)))
    N = max(N, 6)
    W = int(rng.integers(3, 10))
    return family, k, N, hf_days, strat, W


@pytest.mark.parametrize("seed", range(48))
def test_random_shape_and_layout_matches_oracle(native, seed):
    rng = np.random.default_rng(9000 + seed)
    family, k, N, hf_days, strat, W = _case(rng)
    width = k + int(rng.integers(0, 12))                                    # panel wider than the universe: gathered columns
    inp = synthetic.make_kernel_inputs(width, N, W, seed=70000 + seed, hf_days=hf_days)
    n_r, m = inp["n_r"], inp["m"]
    layout = str(rng.choice(["contiguous", "index"])) if width == k else "index"
    conj = strat == "conjugate"
    if layout == "contiguous":
        kw = dict(panel=inp["panel"], start=inp["start"], n_r=n_r)
        if rng.random() < 0.4:
            kw["n_rows"] = rng.integers(max(2, n_r - 7), n_r + 1, W).astype(np.int32)
        if rng.random() < 0.3:
            kw["rf_adj"] = rng.normal(1e-4, 3e-5, size=(W, n_r))
        if conj:
            kw.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], m=m, w0=inp["w0"][:, :k] / inp["w0"][:, :k].sum(1, keepdims=True),
                      n0=inp["n0"])
            if rng.random() < 0.4:
                kw["hf_count"] = rng.integers(max(3, m - 25), m + 1, W).astype(np.int32)
    else:
        cols = np.stack([np.sort(rng.choice(width, k, replace=False)) for _ in range(W)]).astype(np.int32)
        rows = np.stack([inp["start"][w] + np.sort(rng.choice(n_r, n_r, replace=False)) for w in range(W)]).astype(np.int32)
        kw = dict(panel=inp["panel"], start=None, row_idx=rows, col_idx=cols, n_r=n_r,
                  n_rows=rng.integers(max(2, n_r - 7), n_r + 1, W).astype(np.int32))
        if rng.random() < 0.6:
            kw["rf_adj"] = rng.normal(1e-4, 3e-5, size=(W, n_r))
        if conj:
            w0 = np.abs(rng.normal(size=(W, k))) + 0.1
            kw.update(hf_panel=inp["hf_panel"], m=m, w0=w0 / w0.sum(1, keepdims=True), n0=inp["n0"])
            if rng.random() < 0.5:
                kw["hf_row_idx"] = (inp["hf_start"][:, None] + np.arange(m)[None, :]).astype(np.int32)
                kw["hf_count"] = rng.integers(max(3, m - 25), m + 1, W).astype(np.int32)
            else:
                kw["hf_start"] = inp["hf_start"]
    ref, rstat, raux = oracle.posterior_batch_c(strat, k, N, 5.0, **kw)
    got, status, aux = native.posterior_batch(strat, k, N, 5.0, **kw)
    tag = f"seed={seed} {family} k={k} N={N} m={m} {strat} {layout}"
    assert (status == rstat).all(), (tag, status, rstat)
    ok = rstat == 0
    if ok.any():
        np.testing.assert_allclose(got[ok], ref[ok], rtol=0, atol=1e-10, err_msg=tag)
        if conj:
            np.testing.assert_allclose(aux[ok, :6], raux[ok, :6], rtol=1e-11, atol=1e-14, err_msg=tag)
