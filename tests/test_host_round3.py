"""Round 3 host logic that needs no GPU: the `register_strategy` hook, error isolation and the cap
of the cross-spec prefetch, the membership provider in the weights cache key, the lifetime
bookkeeping of `_native` (the process-level test is tests/test_gpu_lifetime.py)."""
import numpy as np
import pandas as pd
import pytest

from incorporating_different_sources_amd import synthetic

from test_host_boundary import _OracleNative, _conj_spec, _spec


@pytest.fixture(scope="module")
def pc():
    from incorporating_different_sources_amd import portfolio_calculations
    return portfolio_calculations


def test_registered_strategy_lets_the_spec_loop_finish(pc):
    """`shrinkage` / `black_litterman` raise NotImplementedError (pypfopt is not in the tree: parity unpinned) unless the
    caller registers a function with the REFERENCE's signature; the dispatch then passes it the frames of ref:999-1011
    and the backtest replays its weights."""
    md, _ = synthetic.make_market_data(n_tickers=9, n_days=70, seed=21)
    days = md["stock_prices_df"].index
    spec = dict(_spec("shrinkage", 5, 20, "daily", "weekly"), risk_aversion=5)
    with pytest.raises(NotImplementedError):
        pc.backtest_portfolio(spec, days[30], days[-1], md)
    with pytest.raises(ValueError):
        pc.register_strategy("jeffreys", lambda *a: None)
    seen = []

    def shrink(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
        # what the reference's function receives: the k largest stocks' prices up to the date, the whole rf frame
        assert k_stock_prices_df.index[-1] == trading_date_ts and k_stock_prices_df.shape[1] == 5
        assert risk_free_rate_df is md["risk_free_rate_df"]
        seen.append(trading_date_ts)
        cols = list(k_stock_prices_df.columns)[::-1]                    # any order: the dispatch aligns by label
        return pd.DataFrame({"Weight": np.linspace(0.1, 0.3, 5)}, index=pd.Index(cols, name="Stock"))

    def bl(portfolio_spec, trading_date_ts, k_stock_market_caps_df, k_stock_prices_df, risk_free_rate_df):
        assert list(k_stock_market_caps_df.columns) == list(k_stock_prices_df.columns)
        return pd.DataFrame({"Weight": np.full(5, 0.2)}, index=pd.Index(k_stock_prices_df.columns, name="Stock"))

    try:
        pc.register_strategy("shrinkage", shrink)
        pc.register_strategy("black_litterman", bl)
        res = pc.backtest_portfolio(spec, days[30], days[-1], md)
        assert len(seen) == len(res["portfolio_weights_metrics_df"]) > 3
        assert np.isfinite(res["portfolio_simple_returns_series"].to_numpy()).all()
        w = pc.calculate_portfolio_weights(pd.Timestamp(days[40]), spec, md)
        vw = pc.calculate_portfolio_weights(pd.Timestamp(days[40]), dict(spec, weighting_strategy="vw"), md)
        assert w.index.equals(vw.index)                                  # cap-descending, as ref:1097 requires
        np.testing.assert_allclose(np.sort(w["Weight"].to_numpy()), np.linspace(0.1, 0.3, 5))
        res_bl = pc.backtest_portfolio(dict(spec, weighting_strategy="black_litterman"), days[30], days[-1], md)
        assert res_bl["portfolio_weights_metrics_df"]["max_long"].eq(0.2).all()
        # the reference-named module functions reach the registered ones too
        assert pc.calculate_shrinkage_portfolio(spec, seen[0], md["stock_prices_df"].iloc[:, :5].loc[:seen[0]],
                                                md["risk_free_rate_df"]).shape == (5, 1)
        both = pc.backtest_portfolios({"a": spec, "b": dict(spec, weighting_strategy="vw")}, days[30], days[-1], md)
        assert set(both) == {"a", "b"}                                    # a registered strategy is not skipped
    finally:
        pc.register_strategy("shrinkage", None)
        pc.register_strategy("black_litterman", None)
    with pytest.raises(NotImplementedError):
        pc.calculate_portfolio_weights(pd.Timestamp(days[40]), spec, md)


def test_prefetch_of_siblings_never_fails_the_spec_itself(pc, monkeypatch):
    """a sibling of the last grid whose market data is missing (no EPU frame) must not fail - or change - a
    VIX spec's own backtest; the sibling's problem surfaces in the sibling's own call."""
    from incorporating_different_sources_amd import batch, portfolio_specs
    md, _ = synthetic.make_market_data(n_tickers=8, n_days=90, seed=14)
    days = md["stock_prices_df"].index
    vix, epu = _conj_spec("conjugate_hf_vix_vw"), _conj_spec("conjugate_hf_epu_vw")
    fake = _OracleNative(1)
    monkeypatch.setattr(pc, "_native", fake)
    monkeypatch.setattr(portfolio_specs, "_LAST_GRID", {"a": vix, "b": epu})
    alone = pc.backtest_portfolio(vix, days[30], days[-1], md)
    batch.clear_panel_cache()
    broken = {key: val for key, val in md.items() if key != "epu_prices_df"}
    fake.calls.clear()
    res = pc.backtest_portfolio(vix, days[30], days[-1], broken)           # the joint batch fails, the spec does not
    assert fake.calls == [len(days) - 30]
    for key in alone:
        assert np.array_equal(alone[key].to_numpy(), res[key].to_numpy(), equal_nan=True)
    with pytest.raises(KeyError):
        pc.backtest_portfolio(epu, days[30], days[-1], broken)
    # the cap: no room for a sibling -> single-spec batches
    batch.clear_panel_cache()
    fake.calls.clear()
    monkeypatch.setattr(pc, "PREFETCH_MAX_WINDOWS", len(days) - 30)
    pc.backtest_portfolio(vix, days[30], days[-1], md)
    assert fake.calls == [len(days) - 30]
    # and the opt-out
    batch.clear_panel_cache()
    fake.calls.clear()
    monkeypatch.setattr(pc, "PREFETCH_MAX_WINDOWS", 1 << 20)
    monkeypatch.setattr(pc, "PREFETCH_SIBLINGS", False)
    pc.backtest_portfolio(vix, days[30], days[-1], md)
    assert fake.calls == [len(days) - 30]
    batch.clear_panel_cache()


def test_weights_cache_key_includes_the_membership_provider(pc, monkeypatch):
    """the same frames with another `index_constituents` provider are another universe: the weights a
    cross-spec batch cached for the first provider must not be returned for the second."""
    from incorporating_different_sources_amd import batch, portfolio_specs
    md, tickers = synthetic.make_market_data(n_tickers=9, n_days=80, seed=15)
    days = [pd.Timestamp(d) for d in md["stock_prices_df"].index[30:]]
    fake = _OracleNative(1)
    monkeypatch.setattr(pc, "_native", fake)
    specs = [_conj_spec("conjugate_hf_vix_vw"), _conj_spec("conjugate_hf_vix_ew")]
    md_a = dict(md, index_constituents=lambda ts: tickers)
    md_b = dict(md, index_constituents=lambda ts: tickers[2:])            # the two largest-index names are out
    pc.calculate_weights_for_specs(days, specs, md_a)                     # fills the cache for provider A
    fake.calls.clear()
    wa, la, _, _ = pc._weights_for_dates(days, specs[0], md_a)
    assert fake.calls == []                                               # served from the cache
    wb, lb, _, _ = pc._weights_for_dates(days, specs[0], md_b)
    assert fake.calls == [len(days)]                                      # NOT served from provider A's entry
    assert all(set(l) <= set(tickers[2:]) for l in lb) and la != lb
    batch.clear_panel_cache()


def test_native_tracks_live_objects_and_finalisers_stand_down():
    """The bookkeeping half of the exit-time fix (no GPU needed): Device registers itself for the ONE atexit hook, and
    once that hook has run (or the interpreter is finalising) `__del__` makes no library call."""
    from incorporating_different_sources_amd import _native
    import atexit
    assert callable(_native.shutdown) and _native._shutdown_at_exit is not None
    calls = []

    class FakeLib:
        def tp_destroy(self, h):
            calls.append(("destroy", h.value))
            return 0

        def tp_batch_destroy(self, b):
            calls.append(("batch_destroy", b.value))
            return 0

    real = _native.lib
    try:
        _native.lib = FakeLib()
        d = _native.Device.__new__(_native.Device)
        d._h = _native.c_void_p(1234)
        import weakref
        d._batches = weakref.WeakSet()
        b = _native.Batch.__new__(_native.Batch)
        b.dev, b._b = d, _native.c_void_p(99)
        d._batches.add(b)
        with _native._live_lock:
            _native._live_devices.add(d)
        _native._closed_for_exit = True            # as after the atexit hook
        b.__del__(); d.__del__()
        assert calls == []                         # finalisers stand down
        _native._closed_for_exit = False
        _native.shutdown()                         # the hook itself: batches first, then the handle
        assert calls == [("batch_destroy", 99), ("destroy", 1234)]
        assert not d._h and not b._b and d not in _native._live_devices
        b.__del__(); d.__del__()
        assert len(calls) == 2
    finally:
        _native.lib = real
        _native._closed_for_exit = False
