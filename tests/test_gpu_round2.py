"""Round-2 additions on the HIP path (-m gpu): the configuration the reference ships, the CHECK identities, cross-spec
batches, the single-process device group, asynchronous uploads from page-locked memory, kept right-hand sides."""
import ctypes
import os

import numpy as np
import pandas as pd
import pytest

from incorporating_different_sources_amd import synthetic

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pc():
    from incorporating_different_sources_amd import portfolio_calculations
    return portfolio_calculations


@pytest.fixture(scope="module")
def native():
    from incorporating_different_sources_amd import _native
    return _native


def test_shipped_configuration_matches_reference(pc):
    """size 50, 250-WEEKLY window, monthly rebalancing (ref portfolio_specs.py:52-62), every spec of
    `create_portfolio_specs()` in main.py's order (src/main.py:48-71) through `backtest_portfolio`; the two pypfopt
    strategies raise NotImplementedError (the reference's unchanged loop would stop at the first of them)."""
    from incorporating_different_sources_amd import portfolio_specs
    g = np.load(os.path.join(GOLDEN, "backtest_shipped_k50_n250_weekly_monthly.npz"))
    md, tickers = synthetic.make_market_data(n_tickers=int(g["n_tickers"]), n_days=int(g["n_days"]),
                                             seed=int(g["seed"]), rf_nan_every=int(g["rf_nan_every"]))
    days = md["stock_prices_df"].index
    ts_start, ts_end = days[int(g["start_idx"])], days[-1]
    specs = portfolio_specs.create_portfolio_specs()
    seen = []
    for name, spec in specs.items():
        strat = spec["weighting_strategy"]
        if strat in ("shrinkage", "black_litterman"):
            with pytest.raises(NotImplementedError):
                pc.backtest_portfolio(spec, ts_start, ts_end, md)
            continue
        np.random.seed(int(g["np_seed"]))                  # Greyserman draws from numpy's global generator
        res = pc.backtest_portfolio(spec, ts_start, ts_end, md)
        seen.append(strat)
        r, t, mdf = (res["portfolio_simple_returns_series"], res["portfolio_turnover_series"],
                     res["portfolio_weights_metrics_df"])
        assert r.name == spec["display_name"]
        assert np.array_equal(mdf.index.values.astype("datetime64[ns]").astype(np.int64), g[f"{strat}_metrics_dates"])
        # Greyserman: the reference's own LU inverse of D_h is that noisy (tests/test_oracle_golden.py)
        rtol, atol = (1e-6, 1e-9) if strat == "greyserman" else (1e-9, 1e-12)
        np.testing.assert_allclose(r.to_numpy(), g[f"{strat}_returns"], rtol=rtol, atol=atol, err_msg=strat)
        np.testing.assert_allclose(t.to_numpy(), g[f"{strat}_turnover"], rtol=rtol, atol=atol, err_msg=strat)
        np.testing.assert_allclose(mdf.to_numpy(), g[f"{strat}_metrics"], rtol=rtol, atol=atol, equal_nan=True, err_msg=strat)
        if strat != "greyserman":
            for j in (0, len(mdf) - 1):
                w = pc.calculate_portfolio_weights(mdf.index[j], spec, md)
                assert [tickers.index(s) for s in w.index] == list(g[f"{strat}_weights_tickers"][j])
                ref = g[f"{strat}_weights"][j]
                np.testing.assert_allclose(w["Weight"].to_numpy(), ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()))
    assert seen == ["vw", "ew", "conjugate_hf_vix_vw", "conjugate_hf_epu_vw", "jeffreys", "jorion", "greyserman"]
    # the whole grid at once: conjugate specs in one device batch, out-of-scope specs skipped
    np.random.seed(int(g["np_seed"]))
    allres = pc.backtest_portfolios({n: s for n, s in specs.items() if s["weighting_strategy"] != "greyserman"},
                                    ts_start, ts_end, md)
    assert len(allres) == 6
    for name, res in allres.items():
        strat = specs[name]["weighting_strategy"]
        np.testing.assert_allclose(res["portfolio_simple_returns_series"].to_numpy(), g[f"{strat}_returns"], rtol=1e-9, atol=1e-12)


def test_check_identities_run_on_device_values(pc):
    """A17: with CHECK on, every helper re-derives its result a second way on the host (ref:81-86, 185-202, 225-242,
    321-330, 420-428) from the matrix the DEVICE produced - and a wrong matrix is caught."""
    g = np.load(os.path.join(GOLDEN, "single_k10_n60.npz"))
    k, N = int(g["k"]), int(g["N"])
    inp = synthetic.make_kernel_inputs(k, N, int(g["W"]), int(g["seed"]), hf_days=int(g["hf_days"]))
    tickers = [f"A{i:04d}" for i in range(k)]
    date, prices_df, intraday_df, caps_df, rf_df = synthetic.window_frames(inp, 0, tickers)
    mcm_df = synthetic.mcm_frame_for_n0(inp["n0"][0], N, prices_df.index)
    spec = {"weighting_strategy": "conjugate_hf_vix_vw", "size": k, "risk_aversion": 5, "turnover_cost": 15,
            "rebalancing_frequency": "daily", "rolling_window": N, "rolling_window_frequency": "daily",
            "mcm_scaling": 1, "display_name": "c"}
    assert pc.CHECK is False
    pc.CHECK = True
    try:
        T = pc.calculate_canonical_statistics_T(spec, date, prices_df, rf_df)
        t = pc.calculate_canonical_statistics_t(spec, date, prices_df, rf_df)
        S0 = pc.calculate_conjugate_prior_S(spec, date, intraday_df, mcm_df)
        c = pc.calculate_conjugate_c(spec, date, prices_df, caps_df, intraday_df, mcm_df)
        w0 = pc.calculate_conjugate_prior_w(spec, date, prices_df, caps_df, mcm_df)
        q0 = pc.calculate_portfolio_variance(w0, S0)
        wts = pc.calculate_conjugate_hf_mcm_portfolio(spec, date, caps_df, prices_df, intraday_df, mcm_df, rf_df)
        np.testing.assert_allclose(T.to_numpy(), g["w0_jeffreys_T"], rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(t.to_numpy().ravel(), g["w0_jeffreys_t"], rtol=1e-11, atol=1e-16)
        assert c == pytest.approx(float(g["w0_conjugate_hf_vix_vw_c"]), rel=1e-12)
        assert q0 == pytest.approx(float(g["w0_conjugate_hf_vix_vw_q0"]), rel=1e-11)
        order = [tickers[i] for i in g["w0_conjugate_hf_vix_vw_order"]]
        np.testing.assert_allclose(wts.loc[order, "Weight"].to_numpy(), g["w0_conjugate_hf_vix_vw_weights"], rtol=0, atol=1e-10)
        # the checks bite: a host second opinion that disagrees with the device's matrix raises the reference's errors
        real = pc._host_excess_returns
        pc._host_excess_returns = lambda *a, **kw: real(*a, **kw) * 1.5
        try:
            with pytest.raises(ValueError, match="Canonical statistics T is not consistent"):
                pc.calculate_canonical_statistics_T(spec, date, prices_df, rf_df)
            with pytest.raises(ValueError, match="Canonical statistics t is not consistent"):
                pc.calculate_canonical_statistics_t(spec, date, prices_df, rf_df)
        finally:
            pc._host_excess_returns = real
    finally:
        pc.CHECK = False


def _conj(strat, k, N, scaling=1, gamma=5):
    return {"weighting_strategy": strat, "size": k, "risk_aversion": gamma, "turnover_cost": 15,
            "rebalancing_frequency": "daily", "rolling_window": N, "rolling_window_frequency": "daily",
            "mcm_scaling": scaling, "display_name": strat}


def test_four_conjugate_specs_in_one_device_batch(pc, native, monkeypatch):
    """VERDICT r1 item 8 on the device: VIX / EPU x vw / ew (+ a second risk aversion) in ONE tp_batch_run equal the
    per-spec results bit for bit - and the per-spec results are the reference's (golden)."""
    g = np.load(os.path.join(GOLDEN, "backtest_k10_n60_daily.npz"))
    md, tickers = synthetic.make_market_data(n_tickers=int(g["n_tickers"]), n_days=int(g["n_days"]),
                                             seed=int(g["seed"]), rf_nan_every=int(g["rf_nan_every"]))
    days = md["stock_prices_df"].index
    dates = [pd.Timestamp(d) for d in days[int(g["start_idx"]):]]
    k, N = int(g["size"]), int(g["N"])
    specs = [_conj("conjugate_hf_vix_vw", k, N), _conj("conjugate_hf_vix_ew", k, N), _conj("conjugate_hf_epu_vw", k, N),
             _conj("conjugate_hf_epu_ew", k, N), _conj("conjugate_hf_epu_vw", k, N, scaling=2, gamma=10)]
    single = [pc._weights_for_dates(dates, sp, md)[0] for sp in specs]
    runs = []
    real = native.posterior_batch
    monkeypatch.setattr(native, "posterior_batch", lambda *a, **kw: (runs.append(len(kw["n_rows"])), real(*a, **kw))[1])
    from incorporating_different_sources_amd import batch
    batch.clear_panel_cache()
    together = pc.calculate_weights_for_specs(dates, specs, md)
    assert runs == [len(specs) * len(dates)]
    for sp, a, b in zip(specs, single, together):
        assert np.array_equal(a, b[0]), sp["weighting_strategy"]
    for sp, w in zip(specs[:4], single[:4]):
        ref = g[f"{sp['weighting_strategy']}_weights"]
        np.testing.assert_allclose(w, ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()))
    batch.clear_panel_cache()


def test_single_process_device_group(pc, native, monkeypatch):
    """The single-process multi-device mode with the devices this box has (one): `run_sharded` through a
    `DeviceGroup` equals the plain batch bit for bit; tp_comm_init_all / tp_group_gather work on a one-rank
    communicator (RCCL world 1)."""
    from incorporating_different_sources_amd import shard
    k, N, W = 20, 40, 300
    inp = synthetic.make_kernel_inputs(k, N, W, seed=99)
    kw = dict(panel=inp["panel"], start=inp["start"], n_r=inp["n_r"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"],
              m=inp["m"], w0=inp["w0"], n0=inp["n0"])
    ref_w, ref_s, ref_aux = native.posterior_batch("conjugate", k, N, 5.0, **kw)
    group = native.DeviceGroup([0])
    w, s, aux = shard.run_sharded(group, "conjugate", k, N, 5.0, kw, want_aux=True)
    assert np.array_equal(w, ref_w) and np.array_equal(s, ref_s) and np.array_equal(aux, ref_aux)
    # the C entry points of the single-process communicator, on one rank
    dev = group.devices[0]
    arr = (ctypes.c_void_p * 1)(dev._h)
    assert native.lib.tp_comm_init_all(arr, 1) == 0
    assert dev.comm_count() == 1
    b = dev.batch("conjugate", k, N, inp["n_r"], 5.0, W, inp["m"])
    b.upload(**{key: val for key, val in kw.items() if key not in ("n_r", "m")})
    b.run()
    wall = np.empty((1, W, k))
    sall = np.empty((1, W), dtype=np.int32)
    barr = (ctypes.c_void_p * 1)(b._b)
    rc = native.lib.tp_group_gather(barr, 1, 0, wall.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                    sall.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert rc == 0, native.lib.tp_last_error(dev._h)
    assert np.array_equal(wall[0], ref_w) and np.array_equal(sall[0], ref_s)
    assert native.lib.tp_comm_init_all(arr, 1) != 0            # already initialised
    b.close()
    group.close()
    # the call surface takes the sharded route when several devices are visible
    monkeypatch.setattr(native, "device_count", lambda: 2)
    g1 = native.DeviceGroup([0])
    monkeypatch.setattr(native, "default_group", lambda: g1)
    monkeypatch.setattr(pc, "SHARD_MIN_WINDOWS", 4)
    kw2 = dict(kw, n_rows=np.full(W, inp["n_r"], np.int32))
    w2, s2, _ = pc._device_posterior_batch("conjugate", k, N, 5.0, kw2)
    assert np.array_equal(w2, ref_w)
    g1.close()


def test_async_upload_from_pinned_memory(native):
    """tp_host_alloc + tp_batch_upload_async: two resident batches, the upload of the next one queued on the copy
    stream while the current one runs; results equal the synchronous path bit for bit."""
    k, N, W = 33, 60, 500
    dev = native.Device(0)
    ins = [synthetic.make_kernel_inputs(k, N, W, seed=500 + i) for i in range(3)]

    def kwargs(inp, pin):
        kw = dict(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
        return {key: (native.pinned_copy(v) if pin else v) for key, v in kw.items()}

    expect = []
    for inp in ins:
        b = dev.batch("conjugate", k, N, inp["n_r"], 5.0, W, inp["m"])
        b.upload(**kwargs(inp, False))
        expect.append(b.run().download(want_aux=False)[0])
        b.close()
    bs = [dev.batch("conjugate", k, N, ins[0]["n_r"], 5.0, W, ins[0]["m"]) for _ in range(2)]
    pinned = [kwargs(inp, True) for inp in ins]
    outs = [(native.pinned_empty((W, k)), native.pinned_empty((W,), np.int32)) for _ in range(2)]
    got = []
    bs[0].upload_async(**pinned[0])
    for i in range(len(ins) * 2):                      # cycle through the inputs twice: buffers are re-uploaded while in use
        if i + 1 < len(ins) * 2:
            bs[(i + 1) % 2].upload_async(**pinned[(i + 1) % len(ins)])
        bs[i % 2].run()
        w, s, _ = bs[i % 2].download(want_aux=False, out=outs[i % 2])
        got.append(w.copy())
    for i, w in enumerate(got):
        assert np.array_equal(w, expect[i % len(ins)]), i
    for b in bs:
        b.upload_wait()
        b.close()
    assert dev.last_timing()["h2d_ms"] > 0
    dev.close()


def test_kept_right_hand_side(native):
    """tp_batch_keep_rhs: the right-hand side comes out of the run that computed the weights - no hidden launch."""
    from oracle import oracle
    k, N, W = 12, 30, 7
    inp = synthetic.make_kernel_inputs(k, N, W, seed=7)
    dev = native.Device(0)
    b = dev.batch("jeffreys", k, N, inp["n_r"], 1.0, W, 0)
    b.upload(panel=inp["panel"], start=inp["start"])
    with pytest.raises(native.TangencyError):
        b.download_rhs()                                  # nothing kept yet: an error, not a launch
    b.keep_rhs()
    with pytest.raises(native.TangencyError):
        b.download_rhs()                                  # kept from the NEXT run on
    b.run()
    t = b.download_rhs()
    for w in range(W):
        X = inp["panel"][inp["start"][w]: inp["start"][w] + inp["n_r"]]
        np.testing.assert_allclose(t[w], oracle.canonical_statistics_t(X), rtol=1e-13, atol=1e-18)
    b.close()
    dev.close()
