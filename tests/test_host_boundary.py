"""Host-side mirror of the reference interface that needs no GPU: spec grid, schedule, universe
selection, passive weightings, turnover - against golden outputs of the unmodified reference."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from incorporating_different_sources_amd import portfolio_specs, synthetic

from conftest import GOLDEN


def test_spec_grid_matches_reference():
    g = json.load(open(os.path.join(GOLDEN, "portfolio_specs.json")))
    specs = portfolio_specs.create_portfolio_specs()
    assert list(specs.keys()) == g["keys"]
    assert specs == g["specs"]
    for key in g["keys"]:
        assert portfolio_specs.get_display_name_from_full_name(key) == g["display"][key]
    for name, color in g["colors"].items():
        assert portfolio_specs.get_color_from_display_name(name) == color
    assert portfolio_specs.get_display_name_from_full_name("nothing") is None
    with pytest.raises(KeyError):
        portfolio_specs.get_color_from_display_name("nothing")


@pytest.fixture(scope="module")
def pc():
    from incorporating_different_sources_amd import portfolio_calculations
    return portfolio_calculations


def _spec(strat, size, N, window_freq, rebal):
    return {"weighting_strategy": strat, "size": size, "risk_aversion": None, "turnover_cost": 15,
            "rebalancing_frequency": rebal, "rolling_window": N, "rolling_window_frequency": window_freq,
            "mcm_scaling": None, "display_name": strat}


@pytest.mark.parametrize("name,strats", [("backtest_k10_n60_daily", ["vw", "ew"]),
                                         ("backtest_k8_n30_weekly_monthly", ["vw"]),
                                         ("backtest_k6_n9_monthly_weekly", ["ew"])])
def test_passive_backtests_match_reference(pc, name, strats):
    """vw / ew need no device: the whole engine (schedule, universe selection, P&L replay, turnover,
    weight metrics) is checked against the reference's outputs on CPU."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    md, tickers = synthetic.make_market_data(n_tickers=int(g["n_tickers"]), n_days=int(g["n_days"]),
                                             seed=int(g["seed"]), rf_nan_every=int(g["rf_nan_every"]))
    days = md["stock_prices_df"].index
    for strat in strats:
        spec = _spec(strat, int(g["size"]), int(g["N"]), str(g["window_freq"]), str(g["rebal"]))
        res = pc.backtest_portfolio(spec, days[int(g["start_idx"])], days[-1], md)
        r, t, mdf = (res["portfolio_simple_returns_series"], res["portfolio_turnover_series"],
                     res["portfolio_weights_metrics_df"])
        assert np.array_equal(mdf.index.values.astype("datetime64[ns]").astype(np.int64), g[f"{strat}_metrics_dates"])
        np.testing.assert_allclose(r.to_numpy(), g[f"{strat}_returns"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(t.to_numpy(), g[f"{strat}_turnover"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(mdf.to_numpy(), g[f"{strat}_metrics"], rtol=1e-12, atol=1e-15, equal_nan=True)
        for j, d in enumerate(mdf.index):
            w = pc.calculate_portfolio_weights(d, spec, md)
            assert w.index.name == "Stock" and list(w.columns) == ["Weight"]
            assert [tickers.index(s) for s in w.index] == list(g[f"{strat}_weights_tickers"][j])
            np.testing.assert_allclose(w["Weight"].to_numpy(), g[f"{strat}_weights"][j], rtol=1e-13)


def test_rebalancing_schedule_rules(pc):
    days = [pd.Timestamp(d) for d in pd.bdate_range("2021-01-04", periods=70)]
    assert pc.rebalancing_schedule(days, "daily") == days
    weekly = pc.rebalancing_schedule(days, "weekly")
    assert weekly[0] == days[0] and all(d.weekday() == 2 for d in weekly[1:])       # Wednesdays (ref:1171)
    monthly = pc.rebalancing_schedule(days, "monthly")
    assert [d.month for d in monthly] == [1, 2, 3, 4]
    # a week without a Wednesday: the ">7 days since the last rebalance" rule fires
    holed = [d for d in days if not (d.weekday() == 2 and d.isocalendar()[1] == 3)]
    w2 = pc.rebalancing_schedule(holed, "weekly")
    assert any(d.weekday() != 2 for d in w2[1:])
    with pytest.raises(ValueError):
        pc.rebalancing_schedule(days, "hourly")


def test_turnover_and_errors(pc):
    a = pd.DataFrame({"Weight": [0.5, 0.3]}, index=["A", "B"])
    b = pd.DataFrame({"Weight": [0.2, 0.6]}, index=["B", "C"])
    # |0.5-0| + |0.3-0.2| + |0-0.6| = 1.2 ; cash |0.8-0.8| = 0 -> 0.6
    assert pc.compute_portfolio_turnover(a, b) == pytest.approx(0.6)
    md, _ = synthetic.make_market_data(n_tickers=6, n_days=40, seed=5)
    d = md["stock_prices_df"].index[10]
    with pytest.raises(ValueError):   # the date must be the last row of the frame (ref:145-147)
        pc.adjust_stock_prices_window(_spec("vw", 4, 5, "daily", "daily"), d, md["stock_prices_df"])
    with pytest.raises(RuntimeError):
        pc.get_k_largest_stocks_market_caps(md["stock_market_caps_df"], md["stock_prices_df"],
                                            md["stock_intraday_prices_df"], d, 4, 5, "hourly")
    caps = md["stock_market_caps_df"].drop(index=d)
    with pytest.raises(ValueError):   # ref:656-658
        pc.get_k_largest_stocks_market_caps(caps, md["stock_prices_df"], md["stock_intraday_prices_df"], d, 4, 5, "daily")


def test_excess_log_returns_quirks(pc):
    """Appendix B-Q2: mean-calendar-gap scaling, label ffill, NaN rate drops the row."""
    days = pd.bdate_range("2021-03-01", periods=8)
    prices = pd.DataFrame({"A": np.linspace(100, 107, 8), "B": np.linspace(50, 57, 8)}, index=days)
    rf = pd.DataFrame({"DTB3": [0.02, 0.02, np.nan, 0.03, 0.03, 0.03, 0.03, 0.03]}, index=days)
    out = pc.calculate_excess_log_returns_from_prices({}, prices, rf)
    assert len(out) == 6 and days[2] not in out.index           # first row (shift) and the NaN-rate row
    gap = pd.Series(days).diff().dt.days.dropna().mean()
    expect = np.log(107 / 106) - ((1 + 0.03) ** (gap / 365) - 1)
    assert out["A"].iloc[-1] == pytest.approx(expect, rel=1e-13)


def test_day_at_a_time_portfolio_equals_array_replay(pc):
    """`Portfolio.update_portfolio` (the reference's day-at-a-time form, ref:1127-1219) and the array replay
    behind `backtest_portfolio` (F2) give the same three outputs."""
    md, tickers = synthetic.make_market_data(n_tickers=12, n_days=140, seed=31, rf_nan_every=13)
    days = md["stock_prices_df"].index
    for strat, rebal in (("vw", "weekly"), ("ew", "daily")):
        spec = _spec(strat, 7, 25, "daily", rebal)
        fast = pc.backtest_portfolio(spec, days[40], days[-1], md)
        dates = [pd.Timestamp(d) for d in days[40:]]
        p = pc.Portfolio(dates[0], spec)
        for ts in dates:
            p.update_portfolio(ts, md)
        slow = (p.get_portfolio_simple_returns(), p.get_portfolio_turnover(), p.get_portfolio_weights_metrics())
        for a, b in zip((fast["portfolio_simple_returns_series"], fast["portfolio_turnover_series"],
                         fast["portfolio_weights_metrics_df"]), slow):
            assert a.index.equals(b.index)
            np.testing.assert_allclose(a.to_numpy(), b.to_numpy(), rtol=1e-12, atol=1e-15, equal_nan=True)
        assert fast["portfolio_simple_returns_series"].name == slow[0].name == strat
        assert list(fast["portfolio_weights_metrics_df"].columns) == list(slow[2].columns)
