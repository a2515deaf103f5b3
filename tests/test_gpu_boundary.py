"""The reference's call surface (portfolio_specs / portfolio_calculations) running on the HIP path,
against golden outputs of the unmodified reference on identical synthetic market data.  -m gpu."""
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


def _spec(strat, size, N, window_freq, rebal):
    simple = strat in ("vw", "ew")
    return {"weighting_strategy": strat, "size": size, "risk_aversion": None if simple else 5, "turnover_cost": 15,
            "rebalancing_frequency": rebal, "rolling_window": N, "rolling_window_frequency": window_freq,
            "mcm_scaling": None if simple or strat in ("jeffreys", "jorion", "greyserman") else 1, "display_name": strat}


@pytest.mark.parametrize("name", ["backtest_k10_n60_daily", "backtest_k8_n30_weekly_monthly",
                                  "backtest_k6_n9_monthly_weekly"])
def test_backtest_portfolio_matches_reference(pc, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    md, tickers = synthetic.make_market_data(n_tickers=int(g["n_tickers"]), n_days=int(g["n_days"]),
                                             seed=int(g["seed"]), rf_nan_every=int(g["rf_nan_every"]))
    days = md["stock_prices_df"].index
    ts_start, ts_end = days[int(g["start_idx"])], days[-1]
    for strat in [str(s) for s in g["strategies"]]:
        spec = _spec(strat, int(g["size"]), int(g["N"]), str(g["window_freq"]), str(g["rebal"]))
        res = pc.backtest_portfolio(spec, ts_start, ts_end, md)
        r, t, mdf = (res["portfolio_simple_returns_series"], res["portfolio_turnover_series"],
                     res["portfolio_weights_metrics_df"])
        assert r.name == strat and t.name == strat
        assert list(mdf.columns) == [str(c) for c in g[f"{strat}_metrics_cols"]]
        assert np.array_equal(r.index.values.astype("datetime64[ns]").astype(np.int64), g[f"{strat}_returns_dates"])
        assert np.array_equal(t.index.values.astype("datetime64[ns]").astype(np.int64), g[f"{strat}_turnover_dates"])
        assert np.array_equal(mdf.index.values.astype("datetime64[ns]").astype(np.int64), g[f"{strat}_metrics_dates"])
        np.testing.assert_allclose(r.to_numpy(), g[f"{strat}_returns"], rtol=1e-10, atol=1e-13, err_msg=strat)
        np.testing.assert_allclose(t.to_numpy(), g[f"{strat}_turnover"], rtol=1e-10, atol=1e-13, err_msg=strat)
        np.testing.assert_allclose(mdf.to_numpy(), g[f"{strat}_metrics"], rtol=1e-10, atol=1e-13, equal_nan=True,
                                   err_msg=strat)
        # per-date weights through the dispatch function, label order included (ref:1097 needs it)
        for i, d in enumerate(mdf.index[:: max(1, len(mdf) // 7)]):
            j = list(mdf.index).index(d)
            w = pc.calculate_portfolio_weights(d, spec, md)
            assert [tickers.index(s) for s in w.index] == list(g[f"{strat}_weights_tickers"][j])
            np.testing.assert_allclose(w["Weight"].to_numpy(), g[f"{strat}_weights"][j], rtol=0, atol=1e-10)


@pytest.mark.parametrize("name", ["single_k10_n60", "single_k33_n80"])
def test_helper_functions_match_reference(pc, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    k, N, W = int(g["k"]), int(g["N"]), int(g["W"])
    inp = synthetic.make_kernel_inputs(k, N, W, int(g["seed"]), hf_days=int(g["hf_days"]))
    tickers = [f"A{i:04d}" for i in range(k)]
    for w in range(W):
        date, prices_df, intraday_df, caps_df, rf_df = synthetic.window_frames(inp, w, tickers)
        mcm_df = synthetic.mcm_frame_for_n0(inp["n0"][w], N, prices_df.index)
        for strat in ("conjugate_hf_vix_vw", "conjugate_hf_vix_ew"):
            spec = _spec(strat, k, N, "daily", "daily")
            tag = f"w{w}_{strat}"
            order = [tickers[i] for i in g[f"{tag}_order"]]
            assert pc.calculate_conjugate_prior_n(spec, date, mcm_df) == pytest.approx(float(g[f"{tag}_n0"]), rel=1e-14)
            S0 = pc.calculate_conjugate_prior_S(spec, date, intraday_df, mcm_df)
            np.testing.assert_allclose(S0.loc[order, order].to_numpy(), g[f"{tag}_S0"], rtol=1e-11, atol=1e-17)
            S1 = pc.calculate_conjugate_posterior_S(spec, date, prices_df, intraday_df, mcm_df, rf_df)
            np.testing.assert_allclose(S1.loc[order, order].to_numpy(), g[f"{tag}_S1"], rtol=1e-11, atol=1e-17)
            c = pc.calculate_conjugate_c(spec, date, prices_df, caps_df, intraday_df, mcm_df)
            assert c == pytest.approx(float(g[f"{tag}_c"]), rel=1e-12)
            w1 = pc.calculate_conjugate_posterior_w(spec, date, prices_df, caps_df, intraday_df, mcm_df, rf_df)
            np.testing.assert_allclose(w1.loc[order, "Weight"].to_numpy(), g[f"{tag}_w1"], rtol=0, atol=1e-10)
            nu = pc.calculate_mean_conjugate_posterior_nu(spec, date, prices_df, caps_df, intraday_df, mcm_df, rf_df)
            np.testing.assert_allclose(nu.loc[order, "Weight"].to_numpy(), g[f"{tag}_nu"], rtol=0, atol=1e-10)
            wts = pc.calculate_conjugate_hf_mcm_portfolio(spec, date, caps_df, prices_df, intraday_df, mcm_df, rf_df)
            np.testing.assert_allclose(wts.loc[order, "Weight"].to_numpy(), g[f"{tag}_weights"], rtol=0, atol=1e-10)
        spec = _spec("jeffreys", k, N, "daily", "daily")
        tag = f"w{w}_jeffreys"
        T = pc.calculate_canonical_statistics_T(spec, date, prices_df, rf_df)
        np.testing.assert_allclose(T.to_numpy(), g[f"{tag}_T"], rtol=1e-12, atol=1e-18)
        t = pc.calculate_canonical_statistics_t(spec, date, prices_df, rf_df)
        np.testing.assert_allclose(t.to_numpy().ravel(), g[f"{tag}_t"], rtol=1e-11, atol=1e-16)
        wts = pc.calculate_jeffreys_portfolio(spec, date, prices_df, rf_df)
        np.testing.assert_allclose(wts["Weight"].to_numpy(), g[f"{tag}_weights"], rtol=1e-10, atol=1e-13)


def test_out_of_scope_strategies_raise(pc):
    md, _ = synthetic.make_market_data(n_tickers=6, n_days=40, seed=5)
    d = md["stock_prices_df"].index[-1]
    for strat in ("shrinkage", "black_litterman"):
        with pytest.raises(NotImplementedError):
            pc.calculate_portfolio_weights(d, _spec(strat, 4, 20, "daily", "daily"), md)
    with pytest.raises(ValueError):
        pc.calculate_portfolio_weights(d, _spec("no_such_strategy", 4, 20, "daily", "daily"), md)


def test_jorion_matches_reference(pc):
    """F3: Jorion's Bayes-Stein portfolio on the device solves vs the reference (single windows)."""
    g = np.load(os.path.join(GOLDEN, "jorion_single.npz"))
    for k, N in ((10, 60), (33, 80), (100, 250)):
        inp = synthetic.make_kernel_inputs(k, N, 2, int(g[f"k{k}_n{N}_seed"]))
        tickers = [f"A{i:04d}" for i in range(k)]
        for w in range(2):
            date, prices_df, intraday_df, caps_df, rf_df = synthetic.window_frames(inp, w, tickers)
            spec = _spec("jorion", k, N, "daily", "daily")
            wts = pc.calculate_jorion_portfolio(spec, date, prices_df, rf_df)
            assert list(wts.index) == tickers and wts.index.name == "Stock"
            np.testing.assert_allclose(wts["Weight"].to_numpy(), g[f"k{k}_n{N}_w{w}_weights"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("name", ["backtest_k10_n60_daily_jorion", "backtest_k8_n30_weekly_monthly_jorion"])
def test_jorion_backtest_matches_reference(pc, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    md, tickers = synthetic.make_market_data(n_tickers=int(g["n_tickers"]), n_days=int(g["n_days"]),
                                             seed=int(g["seed"]), rf_nan_every=int(g["rf_nan_every"]))
    days = md["stock_prices_df"].index
    spec = _spec("jorion", int(g["size"]), int(g["N"]), str(g["window_freq"]), str(g["rebal"]))
    res = pc.backtest_portfolio(spec, days[int(g["start_idx"])], days[-1], md)
    np.testing.assert_allclose(res["portfolio_simple_returns_series"].to_numpy(), g["jorion_returns"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res["portfolio_turnover_series"].to_numpy(), g["jorion_turnover"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res["portfolio_weights_metrics_df"].to_numpy(), g["jorion_metrics"], rtol=1e-9, atol=1e-12,
                               equal_nan=True)


# The reference's own LU inverse of D_h (condition number up to ~1e10) limits how closely ANY implementation
# can agree with its Greyserman weights: see tests/test_oracle_golden.py (GREYSERMAN_RTOL and the
# extended-precision check next to it).
GREYSERMAN_RTOL = 1e-6


def test_greyserman_matches_reference(pc):
    """F3: Greyserman's hierarchical prior, 2 x 1000 ridge solves per window on the device (single windows)."""
    g = np.load(os.path.join(GOLDEN, "greyserman_single.npz"))
    for k, N in ((10, 60), (33, 80), (100, 250)):
        seed = int(g[f"k{k}_n{N}_seed"])
        inp = synthetic.make_kernel_inputs(k, N, 2, seed)
        tickers = [f"A{i:04d}" for i in range(k)]
        for w in range(2):
            date, prices_df, intraday_df, caps_df, rf_df = synthetic.window_frames(inp, w, tickers)
            spec = _spec("greyserman", k, N, "daily", "daily")
            ref = g[f"k{k}_n{N}_w{w}_weights"]
            np.random.seed(seed + w)                      # the reference draws from numpy's global generator
            wts = pc.calculate_greyserman_portfolio(spec, date, prices_df, rf_df)
            assert list(wts.index) == tickers and wts.index.name == "Stock"
            np.testing.assert_allclose(wts["Weight"].to_numpy(), ref, rtol=0, atol=GREYSERMAN_RTOL * np.abs(ref).max())
            again = pc.calculate_greyserman_portfolio(spec, date, prices_df, rf_df,
                                                      draws=(g[f"k{k}_n{N}_w{w}_xi"], g[f"k{k}_n{N}_w{w}_eta"]))
            np.testing.assert_array_equal(again["Weight"].to_numpy(), wts["Weight"].to_numpy())


def test_greyserman_backtest_matches_reference(pc):
    g = np.load(os.path.join(GOLDEN, "backtest_k10_n60_daily_greyserman.npz"))
    md, tickers = synthetic.make_market_data(n_tickers=int(g["n_tickers"]), n_days=int(g["n_days"]),
                                             seed=int(g["seed"]), rf_nan_every=int(g["rf_nan_every"]))
    days = md["stock_prices_df"].index
    spec = _spec("greyserman", int(g["size"]), int(g["N"]), str(g["window_freq"]), str(g["rebal"]))
    np.random.seed(int(g["np_seed"]))
    res = pc.backtest_portfolio(spec, days[int(g["start_idx"])], days[-1], md)
    np.testing.assert_allclose(res["portfolio_simple_returns_series"].to_numpy(), g["greyserman_returns"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(res["portfolio_turnover_series"].to_numpy(), g["greyserman_turnover"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(res["portfolio_weights_metrics_df"].to_numpy(), g["greyserman_metrics"], rtol=1e-6, atol=1e-9,
                               equal_nan=True)
