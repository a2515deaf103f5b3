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
                                         ("backtest_k6_n9_monthly_weekly", ["ew"]),
                                         ("backtest_shipped_k50_n250_weekly_monthly", ["vw", "ew"])])
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


# ---- round 2 -------------------------------------------------------------------------------------------
def test_passive_backtests_need_no_returns_or_risk_free_rate(pc):
    """vw / ew read neither returns nor the risk-free rate in the reference (ref:990-997), so they must
    run from the very first date (no return window yet) and through NaN risk-free rows, like
    `calculate_portfolio_weights` does day by day."""
    md, _ = synthetic.make_market_data(n_tickers=8, n_days=60, seed=11, rf_nan_every=1)    # EVERY rf row is NaN
    md["risk_free_rate_df"].iloc[::2] = 0.02          # mark-to-market needs some finite rates (asof), the window none
    days = md["stock_prices_df"].index
    for strat in ("vw", "ew"):
        spec = _spec(strat, 5, 20, "daily", "weekly")
        res = pc.backtest_portfolio(spec, days[0], days[-1], md)            # starts at days[0]
        assert len(res["portfolio_simple_returns_series"]) == len(days) - 1
        assert res["portfolio_weights_metrics_df"].index[0] == days[0]
        w_first = pc.calculate_portfolio_weights(days[0], spec, md)
        assert len(w_first) == 5 and abs(w_first["Weight"].sum() - 1) < 1e-12
    # a calendar gap larger than mean + 4 days inside the window stops the estimators (ref:44), not vw / ew
    holed = {key: (df.drop(index=days[30:40]) if key != "stock_intraday_prices_df" else df) for key, df in md.items()}
    res = pc.backtest_portfolio(_spec("vw", 5, 20, "daily", "daily"), days[0], days[-1], holed)
    assert len(res["portfolio_simple_returns_series"]) == len(days) - 10 - 1


def test_panel_cache_tracks_every_frame(pc):
    """replacing the VIX frame (same prices frame) must not reuse the old n0 inputs."""
    from incorporating_different_sources_amd import batch
    md, _ = synthetic.make_market_data(n_tickers=8, n_days=80, seed=12)
    days = [pd.Timestamp(d) for d in md["stock_prices_df"].index[40:44]]
    spec = {"weighting_strategy": "conjugate_hf_vix_vw", "size": 5, "risk_aversion": 5, "turnover_cost": 15,
            "rebalancing_frequency": "daily", "rolling_window": 20, "rolling_window_frequency": "daily",
            "mcm_scaling": 1, "display_name": "c"}
    kw1, _ = batch.pack_windows(days, spec, md)
    md2 = dict(md)
    md2["vix_prices_df"] = md["vix_prices_df"] * np.linspace(1.0, 3.0, len(md["vix_prices_df"]))[:, None]
    kw2, _ = batch.pack_windows(days, spec, md2)
    assert not np.allclose(kw1["n0"], kw2["n0"])
    md3 = dict(md)
    md3["stock_market_caps_df"] = md["stock_market_caps_df"].iloc[:, ::-1] * 1.0           # a new caps frame
    assert batch.panels_for(md3, "daily") is not batch.panels_for(md, "daily")
    assert batch.panels_for(md, "daily") is batch.panels_for(md, "daily")
    batch.clear_panel_cache()


class _OracleNative:
    """Stand-in for `_native` on a box without a GPU: the oracle computes (tests may call it), the product's
    packing, sharding and cross-spec batching are what is under test."""

    def __init__(self, n_devices):
        from oracle import oracle
        self.oracle, self.n, self.calls, self.group = oracle, n_devices, [], None
        outer = self

        class Dev:
            def batch(self, strategy, k, N, n_r, gamma, W, m=0, flags=0):
                return outer.Batch(strategy, k, N, n_r, gamma, W, m)

        class Group:
            def __init__(self):
                self.devices = [Dev() for _ in range(outer.n)]
                self.world = outer.n
                self.gathers = 0

            def gather(self, batches, root=0):
                self.gathers += 1
                return np.stack([b.out[0] for b in batches]), np.stack([b.out[1] for b in batches])
        self.Group = Group

    class Batch:
        def __init__(self, strategy, k, N, n_r, gamma, W, m):
            self.a, self.W, self.k = (strategy, k, N, gamma, n_r, m), W, k

        def upload(self, **kw):
            self.kw = kw

        def run(self):
            from oracle import oracle
            strategy, k, N, gamma, n_r, m = self.a
            self.out = oracle.posterior_batch(strategy, k, N, gamma, n_r=n_r, m=m, **self.kw)

        def download(self, want_aux=True):
            return self.out

        def close(self):
            pass

    def device_count(self):
        return self.n

    def default_group(self):
        if self.group is None:
            self.group = self.Group()
        return self.group

    def posterior_batch(self, strategy, k, N, gamma, **kw):
        self.calls.append(len(kw["n_rows"]))
        return self.oracle.posterior_batch(strategy, k, N, gamma, **kw)

    STATUS_NONFINITE, STATUS_NOT_PD = 2, 1


def _conj_spec(strat, scaling=1, gamma=5):
    return {"weighting_strategy": strat, "size": 5, "risk_aversion": gamma, "turnover_cost": 15,
            "rebalancing_frequency": "daily", "rolling_window": 20, "rolling_window_frequency": "daily",
            "mcm_scaling": scaling, "display_name": strat}


def test_weights_shard_over_all_devices_of_the_process(pc, monkeypatch):
    """`backtest_portfolio`'s device batch can use every visible GPU of the ONE process - here two
    stand-in devices; sharded == unsharded bit for bit, one gather.  Since round 3 the route is opt-in
    (`use_device_group` / TP_SHARD=1): without it two visible devices still mean one device."""
    md, _ = synthetic.make_market_data(n_tickers=8, n_days=90, seed=13)
    days = [pd.Timestamp(d) for d in md["stock_prices_df"].index[30:]]
    one = _OracleNative(1)
    monkeypatch.setattr(pc, "_native", one)
    w1, labels1, cols1, caps1 = pc._weights_for_dates(days, _conj_spec("conjugate_hf_vix_vw"), md)
    two = _OracleNative(2)
    monkeypatch.setattr(pc, "_native", two)
    monkeypatch.setattr(pc, "SHARD_MIN_WINDOWS", 8)
    from incorporating_different_sources_amd import batch
    w0, _, _, _ = pc._weights_for_dates(days, _conj_spec("conjugate_hf_vix_vw"), md)     # not opted in: one device
    assert two.calls == [len(days)] and two.default_group().gathers == 0 and np.array_equal(w0, w1)
    two.calls.clear()
    monkeypatch.setattr(pc, "_shard_opt_in", True)
    monkeypatch.setattr(pc, "_shard_group", None)
    w2, labels2, cols2, caps2 = pc._weights_for_dates(days, _conj_spec("conjugate_hf_vix_vw"), md)
    assert np.array_equal(w1, w2) and labels1 == labels2 and np.array_equal(cols1, cols2)
    assert two.default_group().gathers == 1 and two.calls == [] and one.calls == [len(days)]
    res = pc.backtest_portfolio(_conj_spec("jeffreys"), days[0], days[-1], md)
    assert np.isfinite(res["portfolio_simple_returns_series"].to_numpy()).all() and two.default_group().gathers == 2


def test_conjugate_specs_of_a_grid_share_one_device_batch(pc, monkeypatch):
    """four conjugate specs (VIX / EPU x vw / ew, two risk aversions) in ONE device batch equal
    the four per-spec results bit for bit, and the spec loop of main.py (src/main.py:48) finds them cached."""
    md, _ = synthetic.make_market_data(n_tickers=8, n_days=90, seed=14)
    days = md["stock_prices_df"].index
    specs = {f"s{i}": sp for i, sp in enumerate([_conj_spec("conjugate_hf_vix_vw"), _conj_spec("conjugate_hf_epu_vw", 2),
                                                 _conj_spec("conjugate_hf_vix_ew", 1, gamma=10), _conj_spec("conjugate_hf_epu_ew")])}
    fake = _OracleNative(1)
    monkeypatch.setattr(pc, "_native", fake)
    single = {name: pc.backtest_portfolio(sp, days[30], days[-1], md) for name, sp in specs.items()}
    n_dates = len(days) - 30
    assert fake.calls == [n_dates] * 4
    from incorporating_different_sources_amd import batch, portfolio_specs
    batch.clear_panel_cache()
    fake.calls.clear()
    together = pc.backtest_portfolios(specs, days[30], days[-1], md)
    assert fake.calls == [4 * n_dates]                                  # one batch for the four specs
    for name in specs:
        for key in single[name]:
            a, b = single[name][key], together[name][key]
            assert np.array_equal(a.to_numpy(), b.to_numpy(), equal_nan=True), (name, key)
    # the unchanged spec loop: the grid is known from create_portfolio_specs, the first conjugate spec solves its siblings
    batch.clear_panel_cache()
    fake.calls.clear()
    monkeypatch.setattr(portfolio_specs, "_LAST_GRID", specs)
    looped = {name: pc.backtest_portfolio(sp, days[30], days[-1], md) for name, sp in specs.items()}
    assert fake.calls == [4 * n_dates]
    for name in specs:
        assert np.array_equal(looped[name]["portfolio_simple_returns_series"].to_numpy(),
                              single[name]["portfolio_simple_returns_series"].to_numpy())
    monkeypatch.setattr(portfolio_specs, "_LAST_GRID", {})
    batch.clear_panel_cache()
