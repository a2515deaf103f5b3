"""Generate golden vectors by running the UNMODIFIED reference in the build container.

Run:  python oracle/gen_golden.py          (needs /root/reference; writes tests/golden/*.npz)

The reference (`/root/reference/src/portfolio_calculations.py`) is imported from where it lies; none
of its source is copied.  Three third-party / side-effectful modules it imports at module scope but
never uses on the hot path are absent from this image and are registered as inert placeholders before
the import: `pypfopt` (ref:4-8, used only by the out-of-scope shrinkage /
Black-Litterman strategies), `dotenv` (ref:13,18) and `data_handling` (ref:12; its only use on the path,
`extract_unique_tickers` at ref:619, is given the synthetic ticker list).

Inputs are synthetic and regenerated from seeds by `incorporating_different_sources_amd.synthetic`
at test time; fixtures hold the reference's outputs and intermediates (plus inputs for the small
cases and input checksums for the seeded ones).
"""
from __future__ import annotations

import hashlib
import os
import sys
import types

import numpy as np
import pandas as pd

os.environ.setdefault("LOGGING_LEVEL", "WARNING")
sys.dont_write_bytecode = True

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from incorporating_different_sources_amd import synthetic  # noqa: E402

REF_SRC = "/root/reference/src"
OUT = os.path.join(REPO, "tests", "golden")


def import_reference(tickers_holder):
    pp = types.ModuleType("pypfopt")
    for sub in ["EfficientFrontier", "risk_models", "expected_returns", "black_litterman"]:
        m = types.ModuleType("pypfopt." + sub)
        setattr(pp, sub, m)
        sys.modules["pypfopt." + sub] = m
    sys.modules["pypfopt"] = pp
    sys.modules["pypfopt.black_litterman"].BlackLittermanModel = object
    de = types.ModuleType("dotenv")
    de.load_dotenv = lambda *a, **k: None
    sys.modules["dotenv"] = de
    dh = types.ModuleType("data_handling")
    dh.extract_unique_tickers = lambda a, b: list(tickers_holder["tickers"])
    sys.modules["data_handling"] = dh
    sys.path.insert(0, REF_SRC)
    import portfolio_calculations as pc  # the reference module, unmodified
    import portfolio_specs as ps
    return pc, ps


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


window_frames = synthetic.window_frames


def gen_single_windows(pc, name, k, N, hf_days, W, seed, strategies, store_inputs, window_freq="daily"):
    inp = synthetic.make_kernel_inputs(k, N, W, seed, hf_days=hf_days)
    tickers = [f"A{i:04d}" for i in range(k)]
    out = dict(k=k, N=N, hf_days=hf_days, W=W, seed=seed, gamma=5.0,
               panel_sha=sha(inp["panel"]), hf_panel_sha=sha(inp["hf_panel"]), w0_sha=sha(inp["w0"]))
    for w in range(W):
        date, prices_df, intraday_df, caps_df, rf_df = window_frames(inp, w, tickers)
        for strat in strategies:
            spec = {"weighting_strategy": strat, "size": k, "risk_aversion": 5, "turnover_cost": 15,
                    "rebalancing_frequency": "daily", "rolling_window": N,
                    "rolling_window_frequency": window_freq, "mcm_scaling": 1, "display_name": strat}
            tag = f"w{w}_{strat}"
            X = pc.calculate_excess_log_returns_from_prices(
                spec, pc.adjust_stock_prices_window(spec, date, prices_df), rf_df)
            if strat == "jeffreys":
                wts = pc.calculate_jeffreys_portfolio(spec, date, prices_df, rf_df)
                out[f"{tag}_weights"] = wts["Weight"].to_numpy()
                if store_inputs:
                    out[f"w{w}_X"] = X.to_numpy()
                    out[f"{tag}_T"] = pc.calculate_canonical_statistics_T(spec, date, prices_df, rf_df).to_numpy()
                    out[f"{tag}_t"] = pc.calculate_canonical_statistics_t(spec, date, prices_df, rf_df).to_numpy().ravel()
                continue
            # conjugate: the MCM frame is built so that the reference's own n0 equals inp["n0"][w]
            # exactly: a window of N-1 ones and a last value v has avg = (N-1+v)/N and, for v > 1,
            # frac = v/avg -> choose v from the target frac, then store the reference's n0.
            mcm_df = synthetic.mcm_frame_for_n0(inp["n0"][w], N, prices_df.index)
            n0 = pc.calculate_conjugate_prior_n(spec, date, mcm_df)
            S0 = pc.calculate_conjugate_prior_S(spec, date, intraday_df, mcm_df)
            w0 = pc.calculate_conjugate_prior_w(spec, date, prices_df, caps_df, mcm_df)
            c = pc.calculate_conjugate_c(spec, date, prices_df, caps_df, intraday_df, mcm_df)
            S1 = pc.calculate_conjugate_posterior_S(spec, date, prices_df, intraday_df, mcm_df, rf_df)
            w1 = pc.calculate_conjugate_posterior_w(spec, date, prices_df, caps_df, intraday_df, mcm_df, rf_df)
            nu = pc.calculate_mean_conjugate_posterior_nu(spec, date, prices_df, caps_df, intraday_df, mcm_df, rf_df)
            wts = pc.calculate_conjugate_hf_mcm_portfolio(spec, date, caps_df, prices_df, intraday_df, mcm_df, rf_df)
            order = list(w0.index)                      # market-cap descending / column order
            out[f"{tag}_n0"] = float(n0)
            out[f"{tag}_c"] = float(c)
            out[f"{tag}_order"] = np.array([tickers.index(s) for s in order], dtype=np.int32)
            out[f"{tag}_w0"] = w0["Weight"].to_numpy()
            out[f"{tag}_w1"] = w1.loc[order, "Weight"].to_numpy()
            out[f"{tag}_nu"] = nu.loc[order, "Weight"].to_numpy()
            out[f"{tag}_weights"] = wts.loc[order, "Weight"].to_numpy()
            out[f"{tag}_q1"] = float(pc.calculate_portfolio_variance(w1, S1))
            out[f"{tag}_q0"] = float(pc.calculate_portfolio_variance(w0, S0))
            if store_inputs:
                Y = np.log(intraday_df / intraday_df.shift(1)).dropna()
                out[f"w{w}_X"] = X.to_numpy()
                out[f"w{w}_Y"] = Y.to_numpy()
                out[f"{tag}_S0"] = S0.loc[order, order].to_numpy()
                out[f"{tag}_S1"] = S1.loc[order, order].to_numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_backtest(pc, holder, name, strategies, size, N, window_freq, rebal, n_days, n_tickers, seed,
                 start_idx, rf_nan_every=0, np_seed=None):
    md, tickers = synthetic.make_market_data(n_tickers=n_tickers, n_days=n_days, seed=seed,
                                             rf_nan_every=rf_nan_every)
    holder["tickers"] = tickers
    days = md["stock_prices_df"].index
    ts_start, ts_end = days[start_idx], days[-1]
    out = dict(size=size, N=N, n_days=n_days, n_tickers=n_tickers, seed=seed, start_idx=start_idx,
               window_freq=window_freq, rebal=rebal, rf_nan_every=rf_nan_every,
               strategies=np.array(strategies), prices_sha=sha(md["stock_prices_df"].to_numpy()),
               intraday_sha=sha(md["stock_intraday_prices_df"].to_numpy()))
    for strat in strategies:
        simple = strat in ("vw", "ew")
        spec = {"weighting_strategy": strat, "size": size, "risk_aversion": None if simple else 5,
                "turnover_cost": 15, "rebalancing_frequency": rebal, "rolling_window": N,
                "rolling_window_frequency": window_freq, "mcm_scaling": None if simple or strat == "jeffreys" else 1,
                "display_name": strat}
        if np_seed is not None:             # strategies that draw from numpy's global generator (ref:926-927)
            np.random.seed(np_seed)
            out["np_seed"] = np_seed
        res = pc.backtest_portfolio(spec, ts_start, ts_end, md)
        r = res["portfolio_simple_returns_series"]
        t = res["portfolio_turnover_series"]
        mdf = res["portfolio_weights_metrics_df"]
        out[f"{strat}_returns"] = r.to_numpy()
        out[f"{strat}_returns_dates"] = r.index.values.astype("datetime64[ns]").astype(np.int64)
        out[f"{strat}_turnover"] = t.to_numpy()
        out[f"{strat}_turnover_dates"] = t.index.values.astype("datetime64[ns]").astype(np.int64)
        out[f"{strat}_metrics"] = mdf.to_numpy()
        out[f"{strat}_metrics_dates"] = mdf.index.values.astype("datetime64[ns]").astype(np.int64)
        out[f"{strat}_metrics_cols"] = np.array(list(mdf.columns))
        # per-rebalance weights straight from the dispatch function (ref:941)
        wl, tl = [], []
        if np_seed is not None:             # same draw sequence as the backtest above: dates in the same order
            np.random.seed(np_seed)
        for d in mdf.index:
            wdf = pc.calculate_portfolio_weights(d, spec, md)
            wl.append(wdf["Weight"].to_numpy())
            tl.append([tickers.index(s) for s in wdf.index])
        out[f"{strat}_weights"] = np.array(wl)
        out[f"{strat}_weights_tickers"] = np.array(tl, dtype=np.int32)
        print(f"  {name}:{strat}: {len(mdf)} rebalances, {len(r)} returns")
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_result_csv(pc, holder, name, strategies, size, N, window_freq, rebal, n_days, n_tickers, seed, start_idx):
    """The on-disk result formats: the TEXT of the three files src/main.py:79-81 writes for a
    backtest - `Series.to_csv(header=True)` twice and `DataFrame.to_csv(header=True)` - produced by the unmodified
    reference's results.  Stored as data (strategy -> three strings) in tests/golden/<name>_csv.json."""
    import json
    md, tickers = synthetic.make_market_data(n_tickers=n_tickers, n_days=n_days, seed=seed)
    holder["tickers"] = tickers
    days = md["stock_prices_df"].index
    out = dict(size=size, N=N, n_days=n_days, n_tickers=n_tickers, seed=seed, start_idx=start_idx,
               window_freq=window_freq, rebal=rebal, pandas=pd.__version__, files={})
    for strat in strategies:
        simple = strat in ("vw", "ew")
        spec = {"weighting_strategy": strat, "size": size, "risk_aversion": None if simple else 5,
                "turnover_cost": 15, "rebalancing_frequency": rebal, "rolling_window": N,
                "rolling_window_frequency": window_freq, "mcm_scaling": None if simple or strat == "jeffreys" else 1,
                "display_name": "Display " + strat}
        res = pc.backtest_portfolio(spec, days[start_idx], days[-1], md)
        out["files"][strat] = {
            "simple_returns": res["portfolio_simple_returns_series"].to_csv(header=True),          # src/main.py:79
            "turnover": res["portfolio_turnover_series"].to_csv(header=True),                      # src/main.py:80
            "portfolio_weights_metrics": res["portfolio_weights_metrics_df"].to_csv(header=True),  # src/main.py:81
        }
        print(f"  {name}_csv:{strat}: {len(out['files'][strat]['simple_returns'])} bytes of returns text")
    path = os.path.join(OUT, name + "_csv.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    holder = {"tickers": []}
    pc, ps = import_reference(holder)
    which = sys.argv[1:] or ["single", "backtest", "specs", "large", "jorion", "greyserman", "shipped", "csv"]
    conj = ["conjugate_hf_vix_vw", "conjugate_hf_vix_ew"]
    if "single" in which:
        # BASELINE config 1 shapes (k=10, N=60), 4 windows, all intermediates + inputs stored
        gen_single_windows(pc, "single_k10_n60", 10, 60, 1, 4, 20240001, conj + ["jeffreys"], True)
        # BASELINE config 2 shapes (k=100, N=250), 2 windows, intermediates + inputs stored
        gen_single_windows(pc, "single_k100_n250", 100, 250, 1, 2, 20240002, conj + ["jeffreys"], True)
        # odd shapes: k a multiple of 16, k just above/below a tile edge, tiny k
        gen_single_windows(pc, "single_k16_n40", 16, 40, 1, 2, 20240016, conj + ["jeffreys"], True)
        gen_single_windows(pc, "single_k33_n80", 33, 80, 1, 2, 20240033, conj + ["jeffreys"], True)
        gen_single_windows(pc, "single_k3_n12", 3, 12, 1, 2, 20240003, conj + ["jeffreys"], True)
    if "large" in which:
        # outputs only; inputs are regenerated from the seed at test time (checksums stored)
        gen_single_windows(pc, "single_k200_n250", 200, 250, 5, 1, 20240200, conj + ["jeffreys"], False)
        gen_single_windows(pc, "single_k500_n250", 500, 250, 5, 1, 20240003, conj, False)
        gen_single_windows(pc, "single_k1000_n500", 1000, 500, 22, 1, 20240005, conj, False)
    if "backtest" in which:
        strategies = ["conjugate_hf_vix_vw", "conjugate_hf_vix_ew", "conjugate_hf_epu_vw",
                      "conjugate_hf_epu_ew", "jeffreys", "vw", "ew"]
        # BASELINE config 1: k=10 of 14, N=60 daily window, 100 daily rebalances
        gen_backtest(pc, holder, "backtest_k10_n60_daily", strategies, 10, 60, "daily", "daily", 165, 14,
                     20240001, 65)
        # weekly rolling window + monthly rebalancing (the reference's shipped combination, ref
        # portfolio_specs.py:58-60, at reduced size), with NaN risk-free days (Appendix B-Q2)
        gen_backtest(pc, holder, "backtest_k8_n30_weekly_monthly", ["conjugate_hf_vix_vw", "jeffreys", "vw"],
                     8, 30, "weekly", "monthly", 260, 12, 20240011, 170, rf_nan_every=17)
        gen_backtest(pc, holder, "backtest_k6_n9_monthly_weekly", ["conjugate_hf_epu_vw", "jeffreys", "ew"],
                     6, 9, "monthly", "weekly", 300, 9, 20240012, 230)
    if "csv" in which:
        # F4: the text of the result files of src/main.py:79-81 for the configs[0] backtest
        gen_result_csv(pc, holder, "backtest_k10_n60_daily", ["conjugate_hf_vix_vw", "jeffreys", "vw"], 10, 60, "daily",
                       "daily", 165, 14, 20240001, 65)
    if "jorion" in which:
        # F3: Jorion's Bayes-Stein portfolio (ref:851-895) on single windows and in a backtest
        out = {}
        for k, N, seed in ((10, 60, 20240001), (33, 80, 20240033), (100, 250, 20240002)):
            inp = synthetic.make_kernel_inputs(k, N, 2, seed)
            tickers = [f"A{i:04d}" for i in range(k)]
            for w in range(2):
                date, prices_df, intraday_df, caps_df, rf_df = window_frames(inp, w, tickers)
                spec = {"weighting_strategy": "jorion", "size": k, "risk_aversion": 5, "rolling_window": N,
                        "rolling_window_frequency": "daily", "rebalancing_frequency": "daily"}
                wts = pc.calculate_jorion_portfolio(spec, date, prices_df, rf_df)
                out[f"k{k}_n{N}_w{w}_weights"] = wts["Weight"].to_numpy()
            out[f"k{k}_n{N}_seed"] = seed
        np.savez_compressed(os.path.join(OUT, "jorion_single.npz"), **out)
        print("wrote jorion_single.npz")
        gen_backtest(pc, holder, "backtest_k10_n60_daily_jorion", ["jorion"], 10, 60, "daily", "daily", 165, 14,
                     20240001, 65)
        gen_backtest(pc, holder, "backtest_k8_n30_weekly_monthly_jorion", ["jorion"], 8, 30, "weekly", "monthly",
                     260, 12, 20240011, 170, rf_nan_every=17)
    if "greyserman" in which:
        # F3: Greyserman et al. hierarchical prior, 1000 hyper-parameter draws per window (ref:897-938).  The
        # reference draws from numpy's global generator: seed it, and keep the draws next to the weights.
        from oracle import oracle as orc
        out = {}
        for k, N, seed in ((10, 60, 20240001), (33, 80, 20240033), (100, 250, 20240002)):
            inp = synthetic.make_kernel_inputs(k, N, 2, seed)
            tickers = [f"A{i:04d}" for i in range(k)]
            for w in range(2):
                date, prices_df, intraday_df, caps_df, rf_df = window_frames(inp, w, tickers)
                spec = {"weighting_strategy": "greyserman", "size": k, "risk_aversion": 5, "rolling_window": N,
                        "rolling_window_frequency": "daily", "rebalancing_frequency": "daily"}
                np.random.seed(seed + w)
                wts = pc.calculate_greyserman_portfolio(spec, date, prices_df, rf_df)
                out[f"k{k}_n{N}_w{w}_weights"] = wts["Weight"].to_numpy()
                np.random.seed(seed + w)
                xi, eta = orc.greyserman_draws(1000)
                out[f"k{k}_n{N}_w{w}_xi"] = xi
                out[f"k{k}_n{N}_w{w}_eta"] = eta
            out[f"k{k}_n{N}_seed"] = seed
        np.savez_compressed(os.path.join(OUT, "greyserman_single.npz"), **out)
        print("wrote greyserman_single.npz")
        gen_backtest(pc, holder, "backtest_k10_n60_daily_greyserman", ["greyserman"], 10, 60, "daily", "daily", 85, 14,
                     20240001, 65, np_seed=20240077)
    if "shipped" in which:
        # The configuration the reference actually ships (ref portfolio_specs.py:52-62): size 50, 250-WEEKLY window,
        # monthly rebalancing, gamma 5, 15 bp, mcm_scaling 1 - the seven strategies of its grid that are in scope
        # (shrinkage and Black-Litterman need pypfopt, which is not in this image), in main.py's order, on a
        # synthetic market of 60 tickers x 1,424 business days (250 complete weekly bins before the first date),
        # with NaN risk-free days.  Greyserman draws from numpy's global generator: seeded.
        shipped = [s for s in ps.create_portfolio_specs().values()
                   if s["weighting_strategy"] not in ("shrinkage", "black_litterman")]
        assert all(s["size"] == 50 and s["rolling_window"] == 250 and s["rolling_window_frequency"] == "weekly"
                   and s["rebalancing_frequency"] == "monthly" for s in shipped)
        gen_backtest(pc, holder, "backtest_shipped_k50_n250_weekly_monthly", [s["weighting_strategy"] for s in shipped],
                     50, 250, "weekly", "monthly", 1424, 60, 20240050, 1262, rf_nan_every=23, np_seed=20240051)
    if "specs" in which:
        specs = ps.create_portfolio_specs()
        keys = list(specs.keys())
        import json
        with open(os.path.join(OUT, "portfolio_specs.json"), "w") as f:
            json.dump({"keys": keys, "specs": specs,
                       "display": {k: ps.get_display_name_from_full_name(k) for k in keys},
                       "colors": {specs[k]["display_name"]: ps.get_color_from_display_name(specs[k]["display_name"])
                                  for k in keys}}, f, indent=1)
        print("wrote portfolio_specs.json")


if __name__ == "__main__":
    main()
