"""HIP path (through the C-ABI) against the CPU oracle and the reference's golden vectors.  -m gpu."""
import os

import numpy as np
import pytest

from oracle import oracle
from incorporating_different_sources_amd import synthetic

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

WTOL = dict(rtol=1e-9, atol=1e-10)     # north_star: 1e-10 on the weights (fp64)


@pytest.fixture(scope="module")
def native():
    from incorporating_different_sources_amd import _native
    return _native


def _golden_batch(name, strat):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    k, N, W = int(g["k"]), int(g["N"]), int(g["W"])
    n_r = N - 1
    kw = dict(panel=np.concatenate([g[f"w{w}_X"] for w in range(W)], axis=0),
              start=np.arange(W, dtype=np.int64) * n_r, n_r=n_r)
    if strat != "jeffreys":
        m = g["w0_Y"].shape[0]
        kw.update(hf_panel=np.concatenate([g[f"w{w}_Y"] for w in range(W)], axis=0),
                  hf_start=np.arange(W, dtype=np.int64) * m, m=m,
                  w0=np.stack([g[f"w{w}_{strat}_w0"] for w in range(W)]),
                  n0=np.array([float(g[f"w{w}_{strat}_n0"]) for w in range(W)]),
                  col_idx=np.stack([g[f"w{w}_{strat}_order"] for w in range(W)]).astype(np.int32))
    ref = np.stack([g[f"w{w}_{strat}_weights"] for w in range(W)])
    return g, k, N, kw, ref


SMALL = ["single_k3_n12", "single_k10_n60", "single_k16_n40", "single_k33_n80", "single_k100_n250"]


@pytest.mark.parametrize("name", SMALL)
@pytest.mark.parametrize("strat", ["conjugate_hf_vix_vw", "conjugate_hf_vix_ew", "jeffreys"])
def test_hip_matches_reference_golden(native, name, strat):
    g, k, N, kw, ref = _golden_batch(name, strat)
    s = "jeffreys" if strat == "jeffreys" else "conjugate"
    wts, status, aux = native.posterior_batch(s, k, N, 5.0, **kw)
    assert (status == 0).all(), status
    np.testing.assert_allclose(wts, ref, **WTOL)
    if s == "conjugate":
        W = len(ref)
        np.testing.assert_allclose(aux[:, 2], [float(g[f"w{w}_{strat}_c"]) for w in range(W)], rtol=1e-12)
        np.testing.assert_allclose(aux[:, 3], [float(g[f"w{w}_{strat}_q0"]) for w in range(W)], rtol=1e-11)
        np.testing.assert_allclose(aux[:, 4], [float(g[f"w{w}_{strat}_q1"]) for w in range(W)], rtol=1e-9)


@pytest.mark.parametrize("name", ["single_k10_n60", "single_k100_n250"])
def test_hip_S1_matches_reference(native, name):
    strat = "conjugate_hf_vix_vw"
    g, k, N, kw, ref = _golden_batch(name, strat)
    dev = native.default_device()
    n_r = kw.pop("n_r"); m = kw.pop("m")
    b = native.Batch(dev, "conjugate", k, N, n_r, 5.0, len(ref), m)
    b.upload(**kw)
    for w in range(len(ref)):
        S1 = b.download_S1(w)
        np.testing.assert_allclose(S1, g[f"w{w}_{strat}_S1"], rtol=1e-11, atol=1e-16)
    b.close()


@pytest.mark.parametrize("k,N,hf_days", [(1, 8, 1), (2, 9, 1), (7, 20, 1), (15, 40, 1), (16, 40, 1), (17, 60, 1),
                                          (31, 70, 1), (32, 70, 1), (47, 100, 1), (48, 120, 1), (64, 150, 1),
                                          (95, 200, 1), (96, 250, 1), (100, 250, 1), (111, 250, 1), (112, 250, 2),
                                          (128, 300, 2), (150, 320, 2), (191, 400, 3), (192, 400, 3), (200, 420, 3),
                                          (224, 460, 3), (239, 500, 4)])
@pytest.mark.parametrize("strat", ["conjugate", "jeffreys"])
def test_hip_matches_oracle_all_tile_shapes(native, k, N, hf_days, strat):
    """Every tile count 1..16 of the register-tile kernel, k on / next to tile edges."""
    W = 5
    inp = synthetic.make_kernel_inputs(k, N, W, seed=777000 + k, hf_days=hf_days)
    kw = dict(panel=inp["panel"], start=inp["start"], n_r=inp["n_r"])
    if strat == "conjugate":
        kw.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], m=inp["m"], w0=inp["w0"], n0=inp["n0"])
    ref, rstat, raux = oracle.posterior_batch_c(strat, k, N, 5.0, **kw)
    wts, status, aux = native.posterior_batch(strat, k, N, 5.0, **kw)
    assert (status == rstat).all()
    scale = np.abs(ref).max()
    np.testing.assert_allclose(wts, ref, rtol=1e-8, atol=1e-10 * max(1.0, scale))
    if strat == "conjugate":
        np.testing.assert_allclose(aux[:, :6], raux[:, :6], rtol=1e-9, atol=1e-12)
