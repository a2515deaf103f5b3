"""Deterministic synthetic inputs for the posterior path.

Two levels:

* `make_kernel_inputs` - return panels in the panel+offset layout the C-ABI takes
  (`include/tangency_posterior.h`): one-factor daily excess log-returns, iid intraday log-returns,
  value weights from uniform caps, `n0` from a uniform VIX path.
* `make_market_data` - the `market_data` dict of DataFrames that the reference's
  `backtest_portfolio` / `calculate_portfolio_weights` consume
  (`/root/reference/src/portfolio_calculations.py:945-951`, `:1134`; produced in the reference by
  `src/data_handling.py:282-291`), built from the same return model with prices
  `P = 100 * exp(cumsum(x))`.

Generator: `numpy.random.Generator(PCG64(seed))`, `seed = 20240000 + config_id`.
"""
from __future__ import annotations

import numpy as np

BARS_PER_DAY = 78  # 5-minute bars 09:30..15:55

# (k, N, hf_days, W) of BASELINE.json's configs 1..5; m = hf_days*78 - 1 (SURVEY §8(d) well-posedness)
CONFIGS = {
    1: dict(k=10, N=60, hf_days=1, W=100),
    2: dict(k=100, N=250, hf_days=1, W=10_000),
    3: dict(k=500, N=250, hf_days=5, W=50_000),
    4: dict(k=100, N=250, hf_days=1, W=200_000),
    5: dict(k=1000, N=500, hf_days=22, W=1_000_000),
}


def config_shapes(config_id: int) -> dict:
    c = dict(CONFIGS[config_id])
    c["n_r"] = c["N"] - 1
    c["m"] = c["hf_days"] * BARS_PER_DAY - 1
    c["seed"] = 20240000 + config_id
    return c


def make_kernel_inputs(k: int, N: int, W: int, seed: int, hf_days: int = 1, gamma: float = 5.0,
                       mcm_scaling: float = 1.0, prior: str = "vw", hf_period: int = 0) -> dict:
    """Rolling windows `start_w = w` over a shared daily panel of `D = W + n_r` rows, and
    `hf_start_w = w * 78` over a shared intraday-return panel of `(W + hf_days - 1) * 78` rows
    (window w uses the `m = hf_days*78 - 1` returns that start there).  `hf_period > 0`: the intraday panel
    holds only that many days and window w starts at day `w % hf_period` (multi-day intraday windows at
    BASELINE's largest window counts would otherwise need hundreds of GB of host memory)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_r = N - 1
    m = hf_days * BARS_PER_DAY - 1
    D = W + n_r
    beta = rng.uniform(0.5, 1.5, size=k)
    f = rng.normal(0.0, 0.01, size=D)
    panel = 3e-4 + f[:, None] * beta[None, :] + rng.normal(0.0, 0.01, size=(D, k))
    hf_day_count = min(W, hf_period) if hf_period > 0 else W
    H = (hf_day_count + hf_days - 1) * BARS_PER_DAY
    hf_panel = rng.normal(0.0, 0.001, size=(H, k))
    if prior == "vw":
        caps = rng.uniform(1e9, 1e11, size=(W, k))
        caps = -np.sort(-caps, axis=1)                       # descending, like nlargest (ref:654)
        w0 = caps / caps.sum(axis=1, keepdims=True)          # ref:692-695
    else:
        w0 = np.full((W, k), 1.0 / k)                        # ref:670-672
    vix = rng.uniform(12.0, 30.0, size=D + 1)
    # n0 per window: ref:112 (mean of last N obs incl. today), ref:260-265
    csum = np.concatenate([[0.0], np.cumsum(vix)])
    end = np.arange(W) + n_r + 1                              # window w "today" = panel row w + n_r - 1 -> vix idx w + n_r
    avg = (csum[end + 0] - csum[end - N]) / N
    cur = vix[end - 1]
    frac = np.where(cur > avg, cur / avg, avg / cur)
    n0 = N * frac * mcm_scaling
    return dict(k=k, N=N, n_r=n_r, m=m, W=W, gamma=gamma, seed=seed,
                panel=np.ascontiguousarray(panel), start=np.arange(W, dtype=np.int64),
                hf_panel=np.ascontiguousarray(hf_panel),
                hf_start=(np.arange(W, dtype=np.int64) % hf_day_count) * BARS_PER_DAY,
                w0=np.ascontiguousarray(w0), n0=np.ascontiguousarray(n0))


def make_market_data(n_tickers: int = 14, n_days: int = 165, seed: int = 20240001,
                     start_date: str = "2021-01-04", rf_annual: float = 0.02,
                     rf_nan_every: int = 0):
    """Synthetic `market_data` dict with the keys the reference reads (ref:945-951, :1134).

    Business-day calendar; 78 five-minute bars per day; caps follow prices (shares ~ U(1e7,1e9));
    VIX ~ U(12,30); EPU ~ U(50,300); constant annual risk-free rate (optionally NaN every
    `rf_nan_every`-th day to exercise Appendix B-Q2).
    Returns (market_data, tickers).
    """
    import pandas as pd

    rng = np.random.Generator(np.random.PCG64(seed))
    days = pd.bdate_range(start_date, periods=n_days)
    tickers = [f"T{i:03d}" for i in range(n_tickers)]
    beta = rng.uniform(0.5, 1.5, size=n_tickers)
    f = rng.normal(0.0, 0.01, size=n_days)
    x = 3e-4 + f[:, None] * beta[None, :] + rng.normal(0.0, 0.01, size=(n_days, n_tickers))
    prices = 100.0 * np.exp(np.cumsum(x, axis=0))
    stock_prices_df = pd.DataFrame(prices, index=days, columns=tickers)
    stock_simple_returns_df = stock_prices_df.pct_change()
    shares = rng.uniform(1e7, 1e9, size=n_tickers)
    stock_market_caps_df = stock_prices_df * shares[None, :]

    # intraday: per day 78 bars whose last bar closes at the daily price
    bar_times = [pd.Timedelta(hours=9, minutes=30) + pd.Timedelta(minutes=5 * b) for b in range(BARS_PER_DAY)]
    idx = pd.DatetimeIndex([d + bt for d in days for bt in bar_times])
    y = rng.normal(0.0, 0.001, size=(n_days, BARS_PER_DAY, n_tickers))
    intraday_log = np.cumsum(y, axis=1)
    intraday_log = intraday_log - intraday_log[:, -1:, :]           # last bar = daily close
    intraday = np.exp(intraday_log) * prices[:, None, :]
    stock_intraday_prices_df = pd.DataFrame(intraday.reshape(-1, n_tickers), index=idx, columns=tickers)

    vix_prices_df = pd.DataFrame({"VIX": rng.uniform(12.0, 30.0, size=n_days)}, index=days)
    epu_prices_df = pd.DataFrame({"EPU": rng.uniform(50.0, 300.0, size=n_days)}, index=days)
    rf = np.full(n_days, rf_annual)
    if rf_nan_every:
        rf[rf_nan_every - 1::rf_nan_every] = np.nan
    risk_free_rate_df = pd.DataFrame({"DTB3": rf}, index=days)
    sp = 4000.0 * np.exp(np.cumsum(3e-4 + f))
    sp500_prices_df = pd.DataFrame({"SP500TR": sp}, index=days)
    sp500_simple_returns_df = sp500_prices_df.pct_change()
    market_data = {
        "stock_market_caps_df": stock_market_caps_df,
        "stock_prices_df": stock_prices_df,
        "stock_simple_returns_df": stock_simple_returns_df,
        "stock_intraday_prices_df": stock_intraday_prices_df,
        "vix_prices_df": vix_prices_df,
        "epu_prices_df": epu_prices_df,
        "risk_free_rate_df": risk_free_rate_df,
        "sp500_prices_df": sp500_prices_df,
        "sp500_simple_returns_df": sp500_simple_returns_df,
    }
    return market_data, tickers


def window_frames(inp, w, tickers):
    """DataFrames for window w of a `make_kernel_inputs` dict, the way the reference wants them:
    prices P = 100*exp(cumsum(x)) (N rows), intraday prices (m+1 bars on the trading date), caps
    (so that value weights == w0), rf = 0.  Returns (date, prices, intraday, caps, rf)."""
    import pandas as pd

    k, N, n_r, m = inp["k"], inp["N"], inp["n_r"], inp["m"]
    s = int(inp["start"][w])
    x = inp["panel"][s:s + n_r]
    logp = np.concatenate([np.zeros((1, k)), np.cumsum(x, axis=0)], axis=0)
    days = pd.bdate_range("2020-01-01", periods=N)
    date = days[-1]
    prices_df = pd.DataFrame(100.0 * np.exp(logp), index=days, columns=tickers)
    hs = int(inp["hf_start"][w])
    y = inp["hf_panel"][hs:hs + m]
    logh = np.concatenate([np.zeros((1, k)), np.cumsum(y, axis=0)], axis=0)
    # all m+1 bars are stamped inside the trading date, so that the reference's daily filter
    # (date, date+1d] (ref:310-312) keeps exactly these bars whatever m is
    step = pd.Timedelta(seconds=int(6.5 * 3600 / (m + 1)))
    bar_idx = [date + pd.Timedelta(hours=9, minutes=30) + i * step for i in range(m + 1)]
    intraday_df = pd.DataFrame(50.0 * np.exp(logh), index=pd.DatetimeIndex(bar_idx), columns=tickers)
    caps_df = pd.DataFrame([inp["w0"][w] * 1e12], index=[date], columns=tickers)
    rf_df = pd.DataFrame({"DTB3": np.zeros(N)}, index=days)
    return date, prices_df, intraday_df, caps_df, rf_df


def mcm_frame_for_n0(n0_target, N, index):
    """A market-condition series whose reference statistics (ref:112, 257-265) give frac = n0_target/N:
    N-1 ones and a last value v > 1 have avg = (N-1+v)/N and frac = v/avg."""
    import pandas as pd
    frac = n0_target / N
    v = frac * (N - 1) / (N - frac)
    return pd.DataFrame({"VIX": np.r_[np.ones(N - 1), v]}, index=index)
