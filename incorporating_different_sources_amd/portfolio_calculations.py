"""Backtest engine and Bayesian weight estimators with the reference's call surface
(`/root/reference/src/portfolio_calculations.py`, cited below as ref:LINE), MI355X-native underneath.

What is different from the reference is WHERE the arithmetic runs and HOW OFTEN the host touches
pandas:

* every rebalancing date of a backtest is known up front (the schedule depends only on the date list,
  ref:1166-1176), so `backtest_portfolio` extracts all windows first, sends them to the GPU in ONE
  batched call (`_native.posterior_batch` -> `libtangency.so`, include/tangency_posterior.h) and then
  replays the cheap sequential P&L loop (ref:1127-1219) with the weights already in hand;
* the per-window statistics (T, t, S0, c, S1, w1, q1, nu: ref:163-608) never exist on the host: the
  fused HIP kernel produces the weights directly.  The same-named helper functions below exist for
  API compatibility and run a one-window batch on the device.

There is no CPU fallback for the estimators: without `libtangency.so` and a gfx950 GPU they raise.
The passive weightings (vw / ew, ref:661-701) are host-side, as they are inputs (prior weights w0 and
the comparison portfolio), not part of the accelerated path.  Jorion's Bayes-Stein portfolio (ref:851-895)
reuses the device's scatter + Cholesky solve with two right-hand sides, and so does the Greyserman
hierarchical prior (ref:897-938), one ridge-shifted window per hyper-parameter draw.  Shrinkage and
Black-Litterman (ref:703-817: pypfopt, not in the tree) are outside the scope of this build and raise
NotImplementedError - unless the caller registers the reference's own two functions with
`register_strategy` (a caller who has pypfopt), which the dispatch then calls date by date.
"""
from __future__ import annotations

import logging
import os
from datetime import timedelta

import numpy as np
import pandas as pd

try:  # importable both as a package module and as a top-level `portfolio_calculations` (main.py style)
    from . import _native, batch, portfolio_specs, shard
except ImportError:  # pragma: no cover - top-level import with the package directory on sys.path
    import _native  # type: ignore
    import batch  # type: ignore
    import portfolio_specs  # type: ignore
    import shard  # type: ignore

logging_level = os.environ.get("LOGGING_LEVEL", logging.INFO)
logging.basicConfig(level=logging_level)
logger = logging.getLogger(__name__)

# The reference's CHECK flag (ref:30): with it on, every helper re-derives its result a second way on the host and
# raises the reference's ValueError when the two disagree (ref:81-86 w'Sw, ref:185-202 T against the sum of outer
# products, ref:225-242 t against the sum of rows, ref:321-330 the intraday scatter against Y'Y, ref:420-428 the two
# algebraic forms of c) - here the first way is the DEVICE's matrix, read back with tp_batch_download_matrix.  The
# checks are O(n k^2) Python / numpy work per window, so the flag defaults to off (the reference ships it on).
CHECK = False

_CONJUGATE = ("conjugate_hf_vix_vw", "conjugate_hf_vix_ew", "conjugate_hf_epu_vw", "conjugate_hf_epu_ew")
_OUT_OF_SCOPE = ("shrinkage", "black_litterman")
_RESAMPLE_RULE = {"weekly": "W", "monthly": "ME"}


# ======================================================================================================
# small frequency tables (ref:116-134, 299-308, 628-637)
def get_window_annualization_factor(portfolio_spec):
    return {"daily": 252, "weekly": 52, "monthly": 12}[portfolio_spec["rolling_window_frequency"]]


def get_window_trading_days(portfolio_spec):
    per_period = {"daily": 1, "weekly": 5, "monthly": 22}[portfolio_spec["rolling_window_frequency"]]
    return portfolio_spec["rolling_window"] * per_period


def _calendar_days_of(frequency):
    try:
        return {"daily": 1, "weekly": 7, "monthly": 31}[frequency]
    except KeyError:
        logger.error("Unknown rolling window frequency.")
        raise RuntimeError("Unknown rolling window frequency.")


def _resample_last(df, frequency):
    """ref:149-156 / ref:102-109: last observation per calendar week / month ('daily' = unchanged)."""
    if frequency == "daily":
        return df
    if frequency not in _RESAMPLE_RULE:
        return df
    try:
        return df.resample(_RESAMPLE_RULE[frequency]).last()
    except ValueError:  # pandas < 2.2 spells month-end 'M'
        return df.resample("M" if frequency == "monthly" else "W").last()


def _require_last_date(df, trading_date_ts):
    if trading_date_ts != df.index[-1]:
        logger.error(f"trading_date_ts {trading_date_ts} is not the last date in the DataFrame.")
        raise ValueError(f"trading_date_ts {trading_date_ts} must be the last date in the DataFrame.")


# ======================================================================================================
# window preparation on the host (ref:31-62, 136-161): label logic, no O(n k^2) work
def adjust_stock_prices_window(portfolio_spec, trading_date_ts, k_stock_prices_df):
    """Last `rolling_window` (resampled) price rows ending at the trading date (ref:136-161)."""
    k_stock_prices_df = k_stock_prices_df.sort_index()
    _require_last_date(k_stock_prices_df, trading_date_ts)
    window = _resample_last(k_stock_prices_df, portfolio_spec["rolling_window_frequency"])
    return window.iloc[-portfolio_spec["rolling_window"]:]


def calculate_excess_log_returns_from_prices(portfolio_spec, stock_prices_df, risk_free_rate_df):
    """log(P_t/P_{t-1}) minus the risk-free rate scaled to the mean calendar gap of the window's
    dates; rows with any NaN are dropped (ref:31-62)."""
    log_returns = np.log(stock_prices_df / stock_prices_df.shift(1))
    rf_rows = _rf_adjustment_for(stock_prices_df.index, risk_free_rate_df)
    excess = log_returns - rf_rows[:, None]
    return excess.dropna()


def _rf_adjustment_for(index, risk_free_rate_df):
    """Per-row risk-free adjustment for a window's date labels (ref:40-54): annual simple rate ->
    per-period via the MEAN calendar-day gap / 365, forward-filled on labels (NaN values stay NaN)."""
    gaps = index.to_series().diff().dt.days.dropna()
    mean_gap = gaps.mean()
    assert gaps.max() <= mean_gap + 4, "Unexpected large gap between return dates."
    adjusted = (1 + risk_free_rate_df) ** (mean_gap / 365) - 1
    adjusted.index = risk_free_rate_df.index
    return adjusted.reindex(index, method="ffill").to_numpy().reshape(len(index), -1)[:, 0]


def _daily_window_arrays(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
    """Host packing of one window for the device: raw log-returns (n_r x k), the per-row risk-free
    adjustment (subtracted on the device, ref:57) and the asset labels."""
    window = adjust_stock_prices_window(portfolio_spec, trading_date_ts, k_stock_prices_df)
    values = window.to_numpy(dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        log_returns = np.log(values[1:] / values[:-1])
    rf_rows = _rf_adjustment_for(window.index, risk_free_rate_df)[1:]
    keep = ~np.isnan(log_returns).any(axis=1) & ~np.isnan(rf_rows)     # dropna (ref:60)
    return np.ascontiguousarray(log_returns[keep]), np.ascontiguousarray(rf_rows[keep]), list(window.columns)


def _host_excess_returns(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
    """X (n_r x k) on the host, for the CHECK identities only (the device subtracts the risk-free rate itself)."""
    log_returns, rf_rows, _ = _daily_window_arrays(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)
    return log_returns - rf_rows[:, None]


def _intraday_window_returns(portfolio_spec, trading_date_ts, k_stock_intraday_prices_df):
    """Intraday log-returns of the last period (ref:299-314): bars in (date+1d-Delta, date+1d]."""
    span = _calendar_days_of(portfolio_spec["rolling_window_frequency"])
    lo = trading_date_ts - pd.Timedelta(days=span) + pd.Timedelta(days=1)
    hi = trading_date_ts + pd.Timedelta(days=1)
    idx = k_stock_intraday_prices_df.index
    bars = k_stock_intraday_prices_df[(idx > lo) & (idx <= hi)]
    values = bars.to_numpy(dtype=np.float64)
    if len(values) < 2:
        return np.empty((0, values.shape[1] if values.ndim == 2 else 0)), list(bars.columns)
    with np.errstate(divide="ignore", invalid="ignore"):
        log_returns = np.log(values[1:] / values[:-1])
    keep = ~np.isnan(log_returns).any(axis=1)
    return np.ascontiguousarray(log_returns[keep]), list(bars.columns)


# ======================================================================================================
# prior hyper-parameters that are host scalars (ref:90-114, 247-282)
def calculate_average_mcm_window(portfolio_spec, trading_date_ts, mcm_prices_df):
    mcm_prices_df = mcm_prices_df.sort_index()
    _require_last_date(mcm_prices_df, trading_date_ts)
    window = _resample_last(mcm_prices_df, portfolio_spec["rolling_window_frequency"])
    return window.iloc[-portfolio_spec["rolling_window"]:].mean().item()


def calculate_conjugate_prior_n(portfolio_spec, trading_date_ts, mcm_prices_df):
    """n0 = N * max(cur/avg, avg/cur) * mcm_scaling (ref:247-267, Appendix B-Q5)."""
    average = calculate_average_mcm_window(portfolio_spec, trading_date_ts, mcm_prices_df)
    current = mcm_prices_df.loc[trading_date_ts].item()
    fraction = current / average if current > average else average / current
    return portfolio_spec["rolling_window"] * fraction * portfolio_spec["mcm_scaling"]


def calculate_conjugate_posterior_n(portfolio_spec, trading_date_ts, mcm_prices_df, conjugate_prior_n=None):
    if conjugate_prior_n is None:
        conjugate_prior_n = calculate_conjugate_prior_n(portfolio_spec, trading_date_ts, mcm_prices_df)
    return conjugate_prior_n + portfolio_spec["rolling_window"]


# ======================================================================================================
# passive weightings: host side (prior weights / comparison portfolio), ref:661-701
def calculate_equally_weighted_portfolio(portfolio_spec, k_stock_prices_df):
    n = portfolio_spec["size"]
    out = pd.DataFrame({"Weight": [1 / n] * n}, index=k_stock_prices_df.columns)
    out.index.name = "Stock"
    return out


def calculate_value_weighted_portfolio(portfolio_spec, trading_date_ts, k_stock_market_caps_df):
    caps = k_stock_market_caps_df.iloc[-1].sort_values(ascending=False)
    assert k_stock_market_caps_df.index[-1] == trading_date_ts, "The last index date does not match the trading date."
    out = pd.DataFrame(caps / caps.sum())
    out.index.name = "Stock"
    out.columns = ["Weight"]
    return out


def calculate_conjugate_prior_w(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                                mcm_prices_df):
    strategy = portfolio_spec["weighting_strategy"]
    if "vw" in strategy:
        return calculate_value_weighted_portfolio(portfolio_spec, trading_date_ts, k_stock_market_caps_df)
    if "ew" in strategy:
        return calculate_equally_weighted_portfolio(portfolio_spec, k_stock_prices_df)
    logger.error("Unknown conjugate portfolio prior weights.")
    raise ValueError("Unknown conjugate portfolio prior weights.")


def calculate_portfolio_variance(portfolio_weights_df, covariance_matrix_df):
    """w'Sw with label alignment (ref:64-88).  API-compatibility helper: the fused kernel forms q0
    and q1 itself and never calls this."""
    w = portfolio_weights_df.sort_index()
    S = covariance_matrix_df.loc[w.index, w.index].to_numpy()
    v = w["Weight"].to_numpy()
    variance = float(v @ (S @ v))
    if CHECK:  # ref:81-86: the label-aligned pandas product of the unsorted frames gives the same number
        check = portfolio_weights_df["Weight"].T.dot(covariance_matrix_df.dot(portfolio_weights_df["Weight"]))
        if not np.isclose(variance, check, atol=1e-4):
            raise ValueError("Portfolio variance is not consistent.")
    return variance


# ======================================================================================================
# one window on the device
class _WindowOnDevice:
    """One window packed from the reference-style frames and resident on the GPU: the estimator
    helpers below pull whichever quantity they are named after out of it."""

    def __init__(self, portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df,
                 k_stock_market_caps_df=None, k_stock_intraday_prices_df=None, mcm_prices_df=None,
                 conjugate_prior_n=None, conjugate_prior_w_df=None, strategy=None):
        self.spec = portfolio_spec
        X, rf_rows, labels = _daily_window_arrays(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)
        self.labels = labels
        k = len(labels)
        self.k = k
        N = portfolio_spec["rolling_window"]
        conj = (strategy or portfolio_spec["weighting_strategy"]) != "jeffreys"
        gamma = portfolio_spec.get("risk_aversion") or 1.0
        kw = dict(panel=X, start=np.zeros(1, np.int64), rf_adj=rf_rows[None, :])
        m = 0
        self.n0 = None
        self.w0_df = None
        if conj:
            Y, hf_labels = _intraday_window_returns(portfolio_spec, trading_date_ts, k_stock_intraday_prices_df)
            Y = Y[:, [hf_labels.index(s) for s in labels]] if hf_labels != labels else Y
            m = Y.shape[0]
            if conjugate_prior_n is None:
                conjugate_prior_n = calculate_conjugate_prior_n(portfolio_spec, trading_date_ts, mcm_prices_df)
            if conjugate_prior_w_df is None:
                conjugate_prior_w_df = calculate_conjugate_prior_w(portfolio_spec, trading_date_ts, k_stock_prices_df,
                                                                   k_stock_market_caps_df, mcm_prices_df)
            self.n0 = float(conjugate_prior_n)
            self.w0_df = conjugate_prior_w_df
            w0 = conjugate_prior_w_df["Weight"].reindex(labels).to_numpy(dtype=np.float64)
            kw.update(hf_panel=Y, hf_start=np.zeros(1, np.int64), w0=w0[None, :], n0=np.array([self.n0]))
        if portfolio_spec["size"] != k:
            raise ValueError(f"portfolio_spec['size']={portfolio_spec['size']} but the frame has {k} assets")
        dev = _native.default_device()
        self.batch = _native.Batch(dev, "conjugate" if conj else "jeffreys", k, N, max(X.shape[0], 1), gamma, 1, m)
        self.batch.upload(**kw)

    def weights(self):
        w, status, aux = self.batch.run().download()
        _raise_on_status(status)
        return w[0], aux[0]

    def matrix(self, what):
        M, rhs = self.batch.download_matrix(0, what)
        return M, rhs

    def frame(self, M):
        return pd.DataFrame(M, index=self.labels, columns=self.labels)

    def close(self):
        self.batch.close()


def _raise_on_status(status):
    status = np.asarray(status)
    if (status == _native.STATUS_NONFINITE).any() or (status == _native.STATUS_NOT_PD).any():
        # ref:492-494 raises on NaN weights; a non-positive-definite S1 is reported the same way
        # instead of returning the finite garbage an LU inverse would give (Appendix B-Q8)
        logger.error("conjugate_posterior_w_df contains NaN values.")
        raise ValueError("conjugate_posterior_w_df contains NaN values.")


def _weights_frame(values, labels):
    out = pd.DataFrame({"Weight": np.asarray(values, dtype=np.float64)}, index=pd.Index(labels))
    return out


def calculate_canonical_statistics_T(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
    """T = X'X (ref:163-204) - fp64 MFMA Gram on the device."""
    win = _WindowOnDevice(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df, strategy="jeffreys")
    try:
        T, _ = win.matrix("gram")
        if CHECK:  # ref:185-202: T against the sum of the rows' outer products
            X = _host_excess_returns(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)
            T_check = np.zeros_like(T)
            for xi in X:
                T_check += np.outer(xi, xi)
            if not np.isclose(T, T_check, rtol=1e-4, atol=1e-4).all():
                raise ValueError("Canonical statistics T is not consistent.")
        return win.frame(T)
    finally:
        win.close()


def calculate_canonical_statistics_t(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
    """t = X'1 (ref:206-245), returned as a one-column frame like `Series.to_frame()`."""
    win = _WindowOnDevice(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df, strategy="jeffreys")
    try:
        _, t = win.matrix("gram")
        if CHECK:  # ref:225-242: t against the sum of the rows
            X = _host_excess_returns(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)
            t_check = np.zeros_like(t)
            for xi in X:
                t_check += xi
            if not np.isclose(t, t_check, rtol=1e-4, atol=1e-4).all():
                raise ValueError("Canonical statistics t is not consistent.")
        return pd.DataFrame({0: t}, index=pd.Index(win.labels))
    finally:
        win.close()


def calculate_conjugate_prior_S(portfolio_spec, trading_date_ts, k_stock_intraday_prices_df, mcm_prices_df,
                                conjugate_prior_n=None):
    """S0 = n0 * cov(Y) * len(Y) (ref:285-333)."""
    if conjugate_prior_n is None:
        conjugate_prior_n = calculate_conjugate_prior_n(portfolio_spec, trading_date_ts, mcm_prices_df)
    Y, labels = _intraday_window_returns(portfolio_spec, trading_date_ts, k_stock_intraday_prices_df)
    k = len(labels)
    dev = _native.default_device()
    # a one-row dummy daily window: only the prior phase of the kernel is read back
    b = _native.Batch(dev, "conjugate", k, portfolio_spec["rolling_window"], 1, 1.0, 1, Y.shape[0])
    try:
        b.upload(panel=np.zeros((1, k)), start=np.zeros(1, np.int64), hf_panel=Y, hf_start=np.zeros(1, np.int64),
                 w0=np.full((1, k), 1.0 / k), n0=np.array([float(conjugate_prior_n)]))
        S0, _ = b.download_matrix(0, "prior")
    finally:
        b.close()
    if CHECK:  # ref:321-330: cov(Y) * len(Y) against the uncentred Y'Y (rtol = atol = 1e-3, as in the reference)
        if not np.isclose(S0 / conjugate_prior_n, Y.T @ Y, rtol=1e-3, atol=1e-3).all():
            raise ValueError("Realized covariance matrix is not consistent.")
    return pd.DataFrame(S0, index=labels, columns=labels)


def calculate_conjugate_posterior_S(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_intraday_prices_df,
                                    mcm_prices_df, risk_free_rate_df, conjugate_prior_S_df=None):
    """S1 = S0 + T (ref:335-358)."""
    if conjugate_prior_S_df is not None:
        T = calculate_canonical_statistics_T(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)
        return conjugate_prior_S_df + T
    spec = dict(portfolio_spec)
    win = _WindowOnDevice(spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df,
                          k_stock_intraday_prices_df=k_stock_intraday_prices_df, mcm_prices_df=mcm_prices_df,
                          conjugate_prior_w_df=calculate_equally_weighted_portfolio(spec, k_stock_prices_df),
                          strategy="conjugate")
    try:
        S1, _ = win.matrix("posterior")
        return win.frame(S1)
    finally:
        win.close()


def _conjugate_window(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                      k_stock_intraday_prices_df, mcm_prices_df, risk_free_rate_df, conjugate_prior_n=None,
                      conjugate_prior_w_df=None):
    return _WindowOnDevice(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df,
                           k_stock_market_caps_df=k_stock_market_caps_df,
                           k_stock_intraday_prices_df=k_stock_intraday_prices_df, mcm_prices_df=mcm_prices_df,
                           conjugate_prior_n=conjugate_prior_n, conjugate_prior_w_df=conjugate_prior_w_df,
                           strategy="conjugate")


def calculate_conjugate_c(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                          k_stock_intraday_prices_df, mcm_prices_df, conjugate_prior_n=None,
                          conjugate_prior_S_df=None, conjugate_prior_w_df=None):
    """c = 2 n0 / (a + sqrt(a^2 + 4 n0 w0'S0 w0)), a = n0 + size + 2 (ref:382-430)."""
    if conjugate_prior_n is None:
        conjugate_prior_n = calculate_conjugate_prior_n(portfolio_spec, trading_date_ts, mcm_prices_df)
    if conjugate_prior_w_df is None:
        conjugate_prior_w_df = calculate_conjugate_prior_w(portfolio_spec, trading_date_ts, k_stock_prices_df,
                                                           k_stock_market_caps_df, mcm_prices_df)
    if conjugate_prior_S_df is not None:
        q0 = calculate_portfolio_variance(conjugate_prior_w_df, conjugate_prior_S_df)
        a = conjugate_prior_n + portfolio_spec["size"] + 2
        return (2 * conjugate_prior_n) / (a + (a ** 2 + 4 * conjugate_prior_n * q0) ** (1 / 2))
    Y, labels = _intraday_window_returns(portfolio_spec, trading_date_ts, k_stock_intraday_prices_df)
    k = len(labels)
    dev = _native.default_device()
    b = _native.Batch(dev, "conjugate", portfolio_spec["size"], portfolio_spec["rolling_window"], 1, 1.0, 1, Y.shape[0])
    try:
        w0 = conjugate_prior_w_df["Weight"].reindex(labels).to_numpy(dtype=np.float64)
        b.upload(panel=np.zeros((1, k)), start=np.zeros(1, np.int64), hf_panel=Y, hf_start=np.zeros(1, np.int64),
                 w0=w0[None, :], n0=np.array([float(conjugate_prior_n)]))
        _, _, aux = b.run().download()
    finally:
        b.close()
    c = float(aux[0, 2])
    if CHECK:  # ref:420-428: the two algebraic forms of c agree
        q0 = float(aux[0, 3])
        a = conjugate_prior_n + portfolio_spec["size"] + 2
        c_check = (-a + (a ** 2 + 4 * conjugate_prior_n * q0) ** (1 / 2)) / (2 * q0)
        if not np.isclose(c, c_check, atol=1e-3):
            raise ValueError("Portfolio c is not consistent.")
    return c


def calculate_conjugate_posterior_w(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                                    k_stock_intraday_prices_df, mcm_prices_df, risk_free_rate_df, conjugate_c=None,
                                    conjugate_prior_w_df=None, conjugate_prior_S_df=None,
                                    conjugate_posterior_S_df=None):
    """w1 = S1^-1 (c S0 w0 + t) (ref:432-496) - Cholesky solve on the device, no explicit inverse."""
    win = _conjugate_window(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                            k_stock_intraday_prices_df, mcm_prices_df, risk_free_rate_df,
                            conjugate_prior_w_df=conjugate_prior_w_df)
    try:
        weights, aux = win.weights()
        n1, q1 = aux[1], aux[4]
        gamma = portfolio_spec.get("risk_aversion") or 1.0
        w1 = weights * gamma * (n1 - q1) / (n1 + portfolio_spec["size"] + 2)     # undo ref:572-575, 836
        return _weights_frame(w1, win.labels)
    finally:
        win.close()


def calculate_mean_conjugate_posterior_nu(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                                          k_stock_intraday_prices_df, mcm_prices_df, risk_free_rate_df,
                                          conjugate_c=None, conjugate_prior_n=None, conjugate_posterior_n=None,
                                          conjugate_prior_S_df=None, conjugate_posterior_S_df=None,
                                          conjugate_prior_w_df=None, conjugate_posterior_w_df=None):
    """nu = (n1 + size + 2) w1 / (n1 - w1'S1 w1) (ref:499-577): the posterior MEAN of the weights."""
    win = _conjugate_window(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                            k_stock_intraday_prices_df, mcm_prices_df, risk_free_rate_df,
                            conjugate_prior_n=conjugate_prior_n, conjugate_prior_w_df=conjugate_prior_w_df)
    try:
        weights, _ = win.weights()
        gamma = portfolio_spec.get("risk_aversion") or 1.0
        return _weights_frame(weights * gamma, win.labels)
    finally:
        win.close()


def calculate_mean_jeffreys_posterior_nu(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
    """nu = (T - t t'/N)^-1 t (ref:580-608)."""
    win = _WindowOnDevice(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df, strategy="jeffreys")
    try:
        weights, _ = win.weights()
        gamma = portfolio_spec.get("risk_aversion") or 1.0
        return _weights_frame(weights * gamma, win.labels)
    finally:
        win.close()


def calculate_conjugate_hf_mcm_portfolio(portfolio_spec, trading_date_ts, k_stock_market_caps_df, k_stock_prices_df,
                                         k_stock_intraday_prices_df, mcm_prices_df, risk_free_rate_df):
    """weights = nu / risk_aversion (ref:819-836)."""
    win = _conjugate_window(portfolio_spec, trading_date_ts, k_stock_prices_df, k_stock_market_caps_df,
                            k_stock_intraday_prices_df, mcm_prices_df, risk_free_rate_df)
    try:
        weights, _ = win.weights()
        return _weights_frame(weights, win.labels)
    finally:
        win.close()


def calculate_jeffreys_portfolio(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
    """weights = nu / risk_aversion (ref:838-849)."""
    win = _WindowOnDevice(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df, strategy="jeffreys")
    try:
        weights, _ = win.weights()
        return _weights_frame(weights, win.labels)
    finally:
        win.close()


# Strategies of the reference's dispatch (ref:999-1011) that this build does not compute (pypfopt 1.5.5 is an un-vendored
# dependency: parity unpinned) can be supplied by the caller: name -> function with the REFERENCE's signature,
#   shrinkage        f(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)              ref:703-706
#   black_litterman  f(portfolio_spec, trading_date_ts, k_stock_market_caps_df, k_stock_prices_df,
#                      risk_free_rate_df)                                                                  ref:759-763
# returning the reference's frame (index 'Stock', column 'Weight').
_EXTERNAL_STRATEGIES = {}


def register_strategy(name, fn):
    """Let the unchanged spec loop of src/main.py:48 finish: `register_strategy("shrinkage",
    reference_module.calculate_shrinkage_portfolio)` (likewise "black_litterman") makes `calculate_portfolio_weights`
    / `backtest_portfolio` call `fn` per rebalancing date with the frames the reference's dispatch passes
    (ref:999-1011).  `fn=None` removes the registration (the strategy raises NotImplementedError again)."""
    if name not in _OUT_OF_SCOPE:
        raise ValueError(f"register_strategy: only {_OUT_OF_SCOPE} can be supplied from outside, not {name!r}")
    if fn is None:
        _EXTERNAL_STRATEGIES.pop(name, None)
    elif not callable(fn):
        raise TypeError("register_strategy: fn must be callable")
    else:
        _EXTERNAL_STRATEGIES[name] = fn


def _not_in_scope(name):
    strategy = name[len("calculate_"):-len("_portfolio")]

    def f(*args, **kwargs):
        fn = _EXTERNAL_STRATEGIES.get(strategy)
        if fn is not None:
            return fn(*args, **kwargs)
        raise NotImplementedError(
            f"{name} is outside the scope of this build (SURVEY.md section 2: C9-C12); only the conjugate and "
            "Jeffreys posteriors and the passive weightings are provided.  A caller who has pypfopt can supply the "
            f"reference's function: portfolio_calculations.register_strategy({strategy!r}, fn).")
    f.__name__ = name
    return f


calculate_shrinkage_portfolio = _not_in_scope("calculate_shrinkage_portfolio")
calculate_black_litterman_portfolio = _not_in_scope("calculate_black_litterman_portfolio")


def _external_weights_for_dates(trading_dates, portfolio_spec, market_data):
    """A registered out-of-scope strategy, date by date, with the frames the reference's dispatch slices (ref:953-988,
    999-1011); returns what `_weights_for_dates` returns."""
    strategy = portfolio_spec["weighting_strategy"]
    fn = _EXTERNAL_STRATEGIES[strategy]
    members_of = _members_provider(market_data)
    cols, labels, caps = batch.pack_universes(list(trading_dates), portfolio_spec, market_data, members_of=members_of)
    rf = market_data["risk_free_rate_df"]
    weights = np.full(caps.shape, np.nan)
    for i, ts in enumerate(trading_dates):
        uni = _Universe(ts, portfolio_spec, market_data)
        if strategy == "shrinkage":
            frame = fn(portfolio_spec, ts, uni.prices, rf)
        else:
            frame = fn(portfolio_spec, ts, uni.caps, uni.prices, rf)
        col = frame["Weight"] if "Weight" in getattr(frame, "columns", ()) else frame.iloc[:, 0]
        w = col.reindex(labels[i])
        if w.isna().any():
            raise ValueError(f"{strategy}: the registered function returned no weight for {list(w.index[w.isna()])} at {ts}")
        weights[i] = w.to_numpy(dtype=np.float64)
    return weights, labels, cols, caps


def _jorion_from_solves(x_t, x_one, t, T, N, gamma):
    """Jorion's Bayes-Stein weights (ref:869-893) from two device solves with the centred scatter
    J = (T-1) V_hat:  x_t = J^-1 t (t = X'1 = T mu_hat) and x_one = J^-1 1.  V_bar = T/(T-N-2) V_hat, so
    V_bar^-1 = kappa J^-1 with kappa = (T-N-2)(T-1)/T; V_PJ is V_bar scaled plus a multiple of 11', whose
    inverse follows from V_bar^-1 1 by Sherman-Morrison instead of the reference's second LU inverse.
    Vectorised over windows: x_t, x_one, t are [W x N], T is [W]."""
    T = np.asarray(T, dtype=np.float64)[:, None]
    kappa = (T - N - 2) * (T - 1) / T
    mu = t / T
    a = kappa * x_t / T                    # V_bar^-1 mu_hat
    b = kappa * x_one                      # V_bar^-1 1
    s_bb = b.sum(axis=1, keepdims=True)
    s_ab = a.sum(axis=1, keepdims=True)
    s_aa = (mu * a).sum(axis=1, keepdims=True)
    mu_g = s_ab / s_bb                                                   # ref:882
    q = s_aa - 2 * mu_g * s_ab + mu_g ** 2 * s_bb                        # d' V_bar^-1 d
    lam = (N + 2) / q                                                    # ref:885
    v = (N + 2) / ((N + 2) + T * q)                                      # ref:887
    alpha = 1 + 1 / (T + lam)                                            # ref:888: V_PJ = alpha V_bar + beta 11'
    beta = lam / (T * (T + 1 + lam)) / s_bb
    g = (1 - v) * a + v * mu_g * b                                       # V_bar^-1 mu_PJ (ref:889)
    sol = g / alpha - (beta / alpha ** 2) * b * g.sum(axis=1, keepdims=True) / (1 + (beta / alpha) * s_bb)
    return sol / gamma                                                   # ref:893


def _jorion_batch(kw, gamma, k, N):
    """Two launches over the same resident batch: right-hand side t (default) and right-hand side 1."""
    dev = _native.default_device()
    W = len(kw["n_rows"])
    b = _native.Batch(dev, "jeffreys", k, N, kw["n_r"], 1.0, W, 0, flags=_native.FLAG_CENTER_BY_ROWS)
    try:
        b.upload(**{key: val for key, val in kw.items() if key not in ("n_r", "m")})
        b.keep_rhs()                                       # the first run also leaves t = X'1 (ref:222)
        x_t, status, _ = b.run().download(want_aux=False)
        _raise_on_status(status)
        t = b.download_rhs()
        b.keep_rhs(False)
        b.set_rhs(np.ones((W, k)))
        x_one, status, _ = b.run().download(want_aux=False)
        _raise_on_status(status)
    finally:
        b.close()
    return _jorion_from_solves(x_t, x_one, t, kw["n_rows"], k, gamma)


def calculate_jorion_portfolio(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df):
    """Jorion hyper-parameter portfolio (ref:851-895): the two SPD solves run on the device."""
    X, rf_rows, labels = _daily_window_arrays(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)
    k = len(labels)
    kw = dict(panel=X, start=np.zeros(1, np.int64), n_r=X.shape[0], n_rows=np.array([X.shape[0]], np.int32),
              rf_adj=rf_rows[None, :])
    w = _jorion_batch(kw, portfolio_spec["risk_aversion"], k, portfolio_spec["rolling_window"])
    return pd.DataFrame({"Weight": w[0]}, index=pd.Index(labels, name="Stock"))


# ------------------------------------------------------------------------------------------------------
# Greyserman et al. hierarchical prior (ref:897-938): a Monte-Carlo mean over hyper-parameter draws, every
# draw one SPD system D_h x = a_h of the window's size - a batched-Cholesky workload for the device.
GREYSERMAN_DRAWS = 1000            # ref:925
_GREYSERMAN_DATES_PER_LAUNCH = 32  # dates x draws windows per device batch (index arrays are repeated per draw)


def _greyserman_draws(count):
    """(xi_b, eta_b) in the reference's order from numpy's global generator (ref:926-927), so that
    `numpy.random.seed` reproduces the reference's weights.  scipy's `gamma.rvs(a=1, scale=10)` is
    `standard_gamma(1) * 10` on that same generator."""
    xi = np.empty(count)
    eta = np.empty(count)
    for i in range(count):
        xi[i] = np.random.uniform(-1000, 1000)
        eta[i] = np.random.standard_gamma(1.0) * 10
    return xi, eta


def _greyserman_from_solves(u_t, u_one, t, n, xi, eta, k, gamma):
    """Mean over draws of the weights of ref:928-931 from two device solves per draw with the ridge matrix
    B_b = T + (eta_b/2) I:  u_t = B_b^-1 t and u_one = B_b^-1 1  (T = X'X is (n-1) S + n xbar xbar' of ref:929).

    With g = n + kappa, beta = eta_b/2 + kappa xi_b^2 and a_h = (t + kappa xi_b 1)/g the scale matrix of ref:929
    is D_h = B_b + beta 1 1' - g a_h a_h' = B_b + U C U' with U = [1, t].  Woodbury over that rank-two term,
    K = C^-1 + U' B_b^-1 U, collapses (C^-1 [kappa xi_b, 1]'/g = -e_2) to
        D_h^-1 a_h = (K12 u_one - K11 u_t) / det K,
        K11 = 1/beta + 1'u_one,  K12 = 1'u_t - kappa xi_b/beta,  K22 = t'u_t - n - kappa eta_b/(2 beta),
    in which no two large terms cancel (det K is a sum of two negative terms) - unlike the LU inverse of
    D_h itself, whose condition number reaches 1e10.  Shapes: u_t, u_one [W x B x k]; t [W x k]; n [W];
    xi, eta [W x B]."""
    n = np.asarray(n, dtype=np.float64)[:, None]
    kappa = np.array([round(0.1 * v) for v in n[:, 0]], dtype=np.float64)[:, None]       # ref:920
    beta = eta / 2 + kappa * xi ** 2
    K11 = 1 / beta + u_one.sum(axis=2)
    K12 = u_t.sum(axis=2) - kappa * xi / beta
    K22 = (u_t * t[:, None, :]).sum(axis=2) - n - kappa * eta / (2 * beta)
    det = K11 * K22 - K12 ** 2
    x = (K12 / det)[:, :, None] * u_one - (K11 / det)[:, :, None] * u_t                  # D_h^-1 a_h
    nu_h = k                                                                             # ref:921
    scale = 1 / gamma * (nu_h + n + 1) * (1 - 1 / (nu_h + n - k))                        # ref:931
    return (scale[:, :, None] * x).mean(axis=1)


def _greyserman_batch(kw, gamma, k, N, draws=None):
    """All dates of `kw` (the `batch.pack_windows` layout), GREYSERMAN_DRAWS windows per date."""
    dev = _native.default_device()
    n_rows = np.asarray(kw["n_rows"])
    W = len(n_rows)
    B = GREYSERMAN_DRAWS if draws is None else np.asarray(draws[0]).shape[-1]
    if draws is None:
        pairs = [_greyserman_draws(B) for _ in range(W)]          # date-major, as the reference's day loop draws
        xi = np.array([p[0] for p in pairs])
        eta = np.array([p[1] for p in pairs])
    else:
        xi = np.broadcast_to(np.asarray(draws[0], dtype=np.float64), (W, B))
        eta = np.broadcast_to(np.asarray(draws[1], dtype=np.float64), (W, B))
    per_window = ("start", "row_idx", "n_rows", "col_idx", "rf_adj")
    out = np.empty((W, k))
    for lo in range(0, W, _GREYSERMAN_DATES_PER_LAUNCH):
        hi = min(W, lo + _GREYSERMAN_DATES_PER_LAUNCH)
        sub = {key: (np.repeat(np.asarray(val)[lo:hi], B, axis=0) if key in per_window and val is not None else val)
               for key, val in kw.items() if key not in ("n_r", "m")}
        Wb = (hi - lo) * B
        shift = np.zeros((Wb, 2))
        shift[:, 0] = eta[lo:hi].reshape(-1) / 2                                         # eta_b S_h = eta_b/2 (I + 11')
        b = _native.Batch(dev, "jeffreys", k, N, kw["n_r"], 1.0, Wb, 0, flags=_native.FLAG_NO_CENTER)
        try:
            b.set_shift(shift)
            b.upload(**sub)
            b.keep_rhs()                                   # the first run also leaves t = X'1 (ref:222)
            u_t, status, _ = b.run().download(want_aux=False)
            _raise_on_status(status)
            t = b.download_rhs()
            b.keep_rhs(False)
            b.set_rhs(np.ones((Wb, k)))
            u_one, status, _ = b.run().download(want_aux=False)
            _raise_on_status(status)
        finally:
            b.close()
        out[lo:hi] = _greyserman_from_solves(u_t.reshape(hi - lo, B, k), u_one.reshape(hi - lo, B, k),
                                             t.reshape(hi - lo, B, k)[:, 0, :], n_rows[lo:hi], xi[lo:hi], eta[lo:hi],
                                             k, gamma)
    return out


def calculate_greyserman_portfolio(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df, draws=None):
    """Greyserman portfolio (ref:897-938): 2 x 1000 ridge solves on the device, rank-two algebra on the host.
    `draws=(xi, eta)` replaces the random hyper-parameter draws (testing)."""
    X, rf_rows, labels = _daily_window_arrays(portfolio_spec, trading_date_ts, k_stock_prices_df, risk_free_rate_df)
    k = len(labels)
    kw = dict(panel=X, start=np.zeros(1, np.int64), n_r=X.shape[0], n_rows=np.array([X.shape[0]], np.int32),
              rf_adj=rf_rows[None, :])
    w = _greyserman_batch(kw, portfolio_spec["risk_aversion"], k, portfolio_spec["rolling_window"], draws=draws)
    return pd.DataFrame({"Weight": w[0]}, index=pd.Index(labels, name="Stock"))


# ======================================================================================================
# universe selection and dispatch (ref:611-658, 941-1052): host, reproduced not accelerated
def _index_constituents(trading_date_ts, market_data, stock_prices_df):
    """S&P-500 membership at the date.  The reference asks its data layer (ref:619 ->
    data_handling.extract_unique_tickers); here `market_data["index_constituents"]` may be a callable
    `f(date) -> list`, else the reference's `data_handling` module is used when importable, else every
    column counts as a member (synthetic panels)."""
    provider = market_data.get("index_constituents") if isinstance(market_data, dict) else None
    if callable(provider):
        return list(provider(trading_date_ts))
    try:
        import data_handling  # the reference's data layer, if the caller has it on sys.path
        return list(data_handling.extract_unique_tickers(trading_date_ts, trading_date_ts))
    except Exception:
        return list(stock_prices_df.columns)


def get_k_largest_stocks_market_caps(stock_market_caps_df, stock_prices_df, stock_intraday_prices_df,
                                     trading_date_ts, portfolio_size, rolling_window_days, rolling_window_frequency,
                                     tickers_list=None):
    """The `portfolio_size` largest caps among stocks with a complete price window and intraday data
    (ref:611-658).  Column-vectorised; same eligibility rules and the same nlargest ordering."""
    if tickers_list is None:
        tickers_list = _index_constituents(trading_date_ts, {}, stock_prices_df)
    span = _calendar_days_of(rolling_window_frequency)
    members = set(tickers_list)
    cols = [c for c in stock_prices_df.columns
            if c in members and c in stock_market_caps_df.columns and c in stock_intraday_prices_df.columns]
    present = [t for t in tickers_list if t in stock_market_caps_df.columns]
    logger.info(f"Fraction of tickers missing from stock_market_caps_df: "
                f"{(len(tickers_list) - len(present)) / max(len(tickers_list), 1):.2%}")
    recent = stock_prices_df.loc[:trading_date_ts, cols].tail(rolling_window_days)
    full_window = recent.notna().all(axis=0)
    bars = stock_intraday_prices_df.loc[(trading_date_ts - timedelta(days=span)):(trading_date_ts + timedelta(days=1)), cols]
    has_bars = bars.notna().any(axis=0)
    eligible = [c for c in cols if full_window[c] and has_bars[c]]
    if trading_date_ts not in stock_market_caps_df.index:
        logger.error(f"The trading date {trading_date_ts} does not exist in the market capitalizations data.")
        raise ValueError(f"The trading date {trading_date_ts} does not exist in the market capitalizations data.")
    return stock_market_caps_df.loc[trading_date_ts, eligible].dropna().nlargest(portfolio_size)


class _Universe:
    """Everything `calculate_portfolio_weights` slices out of `market_data` for one date (ref:953-988)."""

    def __init__(self, trading_date_ts, portfolio_spec, market_data):
        prices = market_data["stock_prices_df"]
        caps = market_data["stock_market_caps_df"]
        intraday = market_data["stock_intraday_prices_df"]
        tickers = _index_constituents(trading_date_ts, market_data, prices)
        # ref:960 passes the REBALANCING frequency where the callee expects the window frequency (Q7)
        top = get_k_largest_stocks_market_caps(caps, prices, intraday, trading_date_ts, portfolio_spec["size"],
                                               get_window_trading_days(portfolio_spec),
                                               portfolio_spec["rebalancing_frequency"], tickers_list=tickers)
        self.top = top
        self.caps = caps[top.index.intersection(caps.columns)].loc[:trading_date_ts]
        self.prices = prices[top.index.intersection(prices.columns)].loc[:trading_date_ts]
        end_of_day = pd.Timestamp(trading_date_ts).replace(hour=23, minute=59, second=59)
        hf = intraday[top.index.intersection(intraday.columns)]
        self.intraday = hf.loc[hf.index <= end_of_day]
        if self.prices.tail(get_window_trading_days(portfolio_spec)).isna().any().any():
            logger.error("Found NA values in the filtered stock prices.")
            raise ValueError("The filtered stock prices contain NA values.")


def _mcm_frame(portfolio_spec, trading_date_ts, market_data):
    key = "vix_prices_df" if "_vix_" in portfolio_spec["weighting_strategy"] else "epu_prices_df"
    frame = market_data[key]
    return frame.loc[frame.index <= trading_date_ts]


def _pack_window(trading_date_ts, portfolio_spec, market_data):
    """Host packing of one rebalancing date for the batched device call."""
    uni = _Universe(trading_date_ts, portfolio_spec, market_data)
    strategy = portfolio_spec["weighting_strategy"]
    rf = market_data["risk_free_rate_df"]
    X, rf_rows, labels = _daily_window_arrays(portfolio_spec, trading_date_ts, uni.prices, rf)
    item = dict(labels=labels, X=X, rf=rf_rows)
    if strategy in _CONJUGATE:
        mcm = _mcm_frame(portfolio_spec, trading_date_ts, market_data)
        Y, hf_labels = _intraday_window_returns(portfolio_spec, trading_date_ts, uni.intraday)
        if hf_labels != labels:
            Y = Y[:, [hf_labels.index(s) for s in labels]]
        w0_df = calculate_conjugate_prior_w(portfolio_spec, trading_date_ts, uni.prices, uni.caps, mcm)
        item.update(Y=Y, n0=float(calculate_conjugate_prior_n(portfolio_spec, trading_date_ts, mcm)),
                    w0=w0_df["Weight"].reindex(labels).to_numpy(dtype=np.float64))
    return item


# windows below which one GPU is used even when sharding is on (a launch per device has a fixed cost)
SHARD_MIN_WINDOWS = int(os.environ.get("TP_SHARD_MIN_WINDOWS", "2048"))
# The one-process-all-GPUs route is OPT-IN: `TP_SHARD=1` in the environment at import, or `use_device_group(group)`.
# It has never run on more than one physical GPU (this build's boxes have one; the driver's multi-GPU bench uses one
# PROCESS per GPU, bench.py), so a backtest on a multi-GPU node stays on one device unless asked - ADVICE r2.
_shard_group = None
_shard_opt_in = os.environ.get("TP_SHARD", "") not in ("", "0")


def use_device_group(group=True):
    """Shard the batched solves of `backtest_portfolio` over the devices of `group` (a `_native.DeviceGroup`; True:
    every visible GPU, `_native.default_group()`; None / False: back to one device)."""
    global _shard_group, _shard_opt_in
    if group is None or group is False:
        _shard_group, _shard_opt_in = None, False
    elif group is True:
        _shard_group, _shard_opt_in = None, True
    else:
        _shard_group, _shard_opt_in = group, True


def _device_posterior_batch(strategy, k, N, gamma, kw):
    """The batched device call behind every estimator: one device by default; with sharding opted in (`TP_SHARD=1` /
    `use_device_group`), several visible GPUs and enough windows, the windows shard over the group inside this one
    process (`shard.run_sharded`: contiguous shards, one grouped RCCL gather to device 0; the reference's driver is a
    single process, src/main.py:26)."""
    n_windows = len(kw["n_rows"]) if kw.get("n_rows") is not None else len(kw["start"])
    if _shard_opt_in and n_windows >= SHARD_MIN_WINDOWS:
        group = _shard_group if _shard_group is not None else (_native.default_group() if _native.device_count() > 1 else None)
        if group is not None and group.world > 1:
            return shard.run_sharded(group, strategy, k, N, gamma, kw, want_aux=True)
    return _native.posterior_batch(strategy, k, N, gamma, **kw)


def _dates_key(trading_dates):
    return tuple(pd.Timestamp(ts).value for ts in trading_dates)


def _spec_cache_key(portfolio_spec):
    return tuple(portfolio_spec.get(name) for name in ("weighting_strategy", "size", "risk_aversion", "rebalancing_frequency",
                                                        "rolling_window", "rolling_window_frequency", "mcm_scaling"))


def _cache_slot(portfolio_spec, trading_dates, market_data):
    """(cache dict, key, MCM frame) of a conjugate spec's weights.  The cache lives on the panels object, so it dies
    with the price / caps / intraday / risk-free frames it was computed from; the VIX / EPU frame is part of the key
    and is kept alive by the entry (its `id` cannot be recycled while the entry exists)."""
    frame = market_data["vix_prices_df" if "_vix_" in portfolio_spec["weighting_strategy"] else "epu_prices_df"]
    cache = batch.panels_for(market_data, portfolio_spec["rolling_window_frequency"]).weights_cache
    # the universe also depends on who answers "members of the index at this date": the caller's provider is part of
    # the key and is kept alive by the entry, like the frame
    provider = market_data.get("index_constituents") if isinstance(market_data, dict) else None
    provider = provider if callable(provider) else None
    return cache, (_spec_cache_key(portfolio_spec), _dates_key(trading_dates), id(frame), id(provider)), (frame, provider)


def _weights_for_dates(trading_dates, portfolio_spec, market_data):
    """(weights [W x k], labels, column indices [W x k], market caps [W x k]) for MANY rebalancing dates: one
    host pass (`batch.pack_windows`) and, for the estimators, one device batch."""
    strategy = portfolio_spec["weighting_strategy"]
    if strategy in _OUT_OF_SCOPE:
        if strategy in _EXTERNAL_STRATEGIES:
            return _external_weights_for_dates(trading_dates, portfolio_spec, market_data)
        _not_in_scope(f"calculate_{strategy}_portfolio")()
    if strategy not in _CONJUGATE and strategy not in ("jeffreys", "jorion", "greyserman", "vw", "ew"):
        logger.error("Unknown weights spec.")
        raise ValueError("Unknown weights spec.")
    members_of = _members_provider(market_data)
    k, N, gamma = portfolio_spec["size"], portfolio_spec["rolling_window"], portfolio_spec.get("risk_aversion")
    if strategy in ("vw", "ew"):
        # passive strategies: universe selection only (ref:990-997 reads neither returns nor the risk-free rate)
        cols, labels, caps = batch.pack_universes(list(trading_dates), portfolio_spec, market_data, members_of=members_of)
        if strategy == "vw":
            weights = caps / caps.sum(axis=1, keepdims=True)             # ref:692-695 (already cap-descending)
        else:
            weights = np.full(caps.shape, 1 / k)                         # ref:670-672
        return weights, labels, cols, caps
    if strategy in _CONJUGATE:
        cache, key, owners = _cache_slot(portfolio_spec, trading_dates, market_data)
        hit = cache.get(key)
        if hit is not None and hit[1][0] is owners[0] and hit[1][1] is owners[1]:      # filled by a cross-spec batch
            return hit[0]
    kw, labels, caps = batch.pack_windows(list(trading_dates), portfolio_spec, market_data, members_of=members_of,
                                          return_caps=True)
    cols = kw["col_idx"]
    if strategy == "jorion":
        kw.pop("start", None)
        weights = _jorion_batch(kw, gamma, k, N)
    elif strategy == "greyserman":
        weights = _greyserman_batch(kw, gamma, k, N)
    else:
        conj = strategy in _CONJUGATE
        weights, status, aux = _device_posterior_batch("conjugate" if conj else "jeffreys", k, N, gamma, kw)
        _raise_on_status(status)
    return weights, labels, cols, caps


def _batch_family(portfolio_spec):
    """Conjugate specs with equal (size, window, frequencies) see the same windows: only w0 / n0 / gamma differ."""
    return tuple(portfolio_spec.get(name) for name in ("size", "rebalancing_frequency", "rolling_window", "rolling_window_frequency"))


def calculate_weights_for_specs(trading_dates, portfolio_specs_list, market_data):
    """Weights of SEVERAL conjugate specs of one grid for the same dates with ONE host pack, ONE upload of the
    panels and ONE device batch of len(specs) x len(dates) windows ("vectorised over days
    and over specs"): the specs share the windows (rows, columns, risk-free adjustment, intraday rows) and differ in
    the prior only - w0 (vw / ew, ref:361-380), n0 (VIX / EPU, mcm_scaling, ref:247-267) and gamma.  Returns one
    (weights, labels, cols, caps) per spec, bit-identical to `_weights_for_dates` spec by spec (windows are
    independent), and remembers them for the `backtest_portfolio` calls that follow."""
    specs = list(portfolio_specs_list)
    if not specs:
        return []
    if any(sp["weighting_strategy"] not in _CONJUGATE for sp in specs) or len({_batch_family(sp) for sp in specs}) != 1:
        raise ValueError("calculate_weights_for_specs: conjugate specs of equal size, window and frequencies expected")
    trading_dates = list(trading_dates)
    members_of = _members_provider(market_data)
    first = specs[0]
    kw, labels, caps = batch.pack_windows(trading_dates, first, market_data, members_of=members_of, return_caps=True)
    n_dates, n_specs = len(trading_dates), len(specs)
    k, N = first["size"], first["rolling_window"]
    gammas = [sp["risk_aversion"] for sp in specs]
    same_gamma = all(g == gammas[0] for g in gammas)
    per_window = ("row_idx", "n_rows", "col_idx", "rf_adj", "hf_row_idx", "hf_count")
    big = {key: (np.concatenate([val] * n_specs, axis=0) if key in per_window and val is not None else val)
           for key, val in kw.items()}
    priors = [(kw["w0"], kw["n0"])] + [batch.prior_inputs(trading_dates, sp, market_data, caps) for sp in specs[1:]]
    big["w0"] = np.concatenate([p[0] for p in priors], axis=0)
    big["n0"] = np.concatenate([p[1] for p in priors], axis=0)
    # 1/gamma is the last factor of ref:836: with different risk aversions in one batch the device runs with
    # gamma = 1 and the same multiplication happens here (same rounding: (1/gamma) * x in both places)
    weights, status, _ = _device_posterior_batch("conjugate", k, N, gammas[0] if same_gamma else 1.0, big)
    _raise_on_status(status)
    out = []
    for i, sp in enumerate(specs):
        w = weights[i * n_dates:(i + 1) * n_dates]
        if not same_gamma:
            w = 1.0 / sp["risk_aversion"] * w
        res = (w, labels, kw["col_idx"], caps)
        cache, key, owners = _cache_slot(sp, trading_dates, market_data)
        if len(cache) > 64:
            cache.clear()
        cache[key] = (res, owners)
        out.append(res)
    return out


# Cross-spec prefetch of `backtest_portfolio` (off with TP_PREFETCH_SIBLINGS=0): at most this many windows per batch,
# siblings beyond it wait for their own call
PREFETCH_SIBLINGS = os.environ.get("TP_PREFETCH_SIBLINGS", "1") not in ("0", "")
PREFETCH_MAX_WINDOWS = int(os.environ.get("TP_PREFETCH_MAX_WINDOWS", "65536"))


def _prefetch_siblings(portfolio_spec, rebalance_dates, market_data):
    """`backtest_portfolio` is called once per spec (src/main.py:48): when the spec is a conjugate spec of the grid
    `portfolio_specs.create_portfolio_specs()` returned last, solve its conjugate siblings with the same windows in
    the same device batch, so that their own `backtest_portfolio` calls find the weights ready.

    Only an optimisation, so it never changes what the call itself does: the batch is capped
    (`PREFETCH_MAX_WINDOWS` windows: the index arrays are replicated per spec), and ANY failure of the joint batch - a
    sibling whose market data is missing or whose prior strength raises - is logged and dropped: the spec then runs on
    its own, and the sibling's problem surfaces in the sibling's own `backtest_portfolio`, where the reference's loop
    would raise it."""
    if not PREFETCH_SIBLINGS or portfolio_spec["weighting_strategy"] not in _CONJUGATE:
        return
    cache, key, owners = _cache_slot(portfolio_spec, rebalance_dates, market_data)
    if key in cache and cache[key][1][0] is owners[0] and cache[key][1][1] is owners[1]:
        return
    family = _batch_family(portfolio_spec)
    siblings = [sp for sp in portfolio_specs.last_grid().values()
                if sp["weighting_strategy"] in _CONJUGATE and _batch_family(sp) == family
                and _spec_cache_key(sp) != _spec_cache_key(portfolio_spec) and sp.get("risk_aversion")]
    room = PREFETCH_MAX_WINDOWS // max(len(rebalance_dates), 1) - 1
    siblings = siblings[:max(room, 0)]
    if not siblings:
        return
    try:
        calculate_weights_for_specs(rebalance_dates, [portfolio_spec] + siblings, market_data)
    except Exception as exc:      # noqa: BLE001 - see the docstring: the single-spec path decides what the caller sees
        logger.warning(f"cross-spec prefetch dropped ({type(exc).__name__}: {exc}); solving {portfolio_spec.get('display_name')} alone")


def calculate_portfolio_weights_batch(trading_dates, portfolio_spec, market_data):
    """Weights for MANY rebalancing dates with one device call (the batch-native form of ref:941).

    Host side: `batch.pack_windows` turns the dates into panel + row/column-index arrays without
    per-date pandas work; device side: one `posterior_batch` launch."""
    strategy = portfolio_spec["weighting_strategy"]
    if strategy in ("vw", "ew"):
        return [calculate_portfolio_weights(d, portfolio_spec, market_data) for d in trading_dates]
    if not trading_dates:
        if strategy in _OUT_OF_SCOPE:
            _not_in_scope(f"calculate_{strategy}_portfolio")()
        return []
    weights, labels, _, _ = _weights_for_dates(trading_dates, portfolio_spec, market_data)
    return [pd.DataFrame({"Weight": weights[i]}, index=pd.Index(labels[i], name="Stock")) for i in range(len(labels))]


def _members_provider(market_data):
    """Index membership per date for the batch packer: None = every column (synthetic panels)."""
    provider = market_data.get("index_constituents") if isinstance(market_data, dict) else None
    if callable(provider):
        return provider
    try:
        import data_handling  # the reference's data layer, if the caller has it on sys.path
        return lambda ts: data_handling.extract_unique_tickers(ts, ts)
    except Exception:
        return None


def calculate_portfolio_weights(trading_date_ts, portfolio_spec, market_data):
    """Strategy dispatch for one date (ref:941-1052); rows ordered by market cap, index 'Stock'."""
    strategy = portfolio_spec["weighting_strategy"]
    if strategy == "vw":
        uni = _Universe(trading_date_ts, portfolio_spec, market_data)
        return calculate_value_weighted_portfolio(portfolio_spec, trading_date_ts, uni.caps)
    if strategy == "ew":
        uni = _Universe(trading_date_ts, portfolio_spec, market_data)
        return calculate_equally_weighted_portfolio(portfolio_spec, uni.prices)
    return calculate_portfolio_weights_batch([trading_date_ts], portfolio_spec, market_data)[0]


# ======================================================================================================
# backtest engine (ref:1054-1238)
def compute_portfolio_turnover(portfolio_weights_before_df, portfolio_weights_after_df):
    """Half the L1 change of the weights, the risk-free position included (ref:1054-1075)."""
    both = portfolio_weights_before_df.merge(portfolio_weights_after_df, how="outer", left_index=True,
                                             right_index=True, suffixes=("_before", "_after")).fillna(0)
    traded = (both["Weight_before"] - both["Weight_after"]).abs().sum()
    cash = abs(portfolio_weights_before_df["Weight"].sum() - portfolio_weights_after_df["Weight"].sum())
    return (traded + cash) / 2


def calculate_average_distance_to_comparison_portfolio(portfolio_weights_df, portfolio_spec, trading_date_ts,
                                                       market_data, comparison_portfolio_weighting_strategy):
    """Mean absolute distance to the value-weighted portfolio of the same universe (ref:1077-1104)."""
    if comparison_portfolio_weighting_strategy != "vw":
        raise ValueError("Unknown comparison portfolio.")
    comparison_spec = {"size": portfolio_spec["size"], "rebalancing_frequency": portfolio_spec["rebalancing_frequency"],
                       "rolling_window": portfolio_spec["rolling_window"],
                       "rolling_window_frequency": portfolio_spec["rolling_window_frequency"],
                       "weighting_strategy": "vw"}
    comparison = calculate_portfolio_weights(trading_date_ts, comparison_spec, market_data)
    if not portfolio_weights_df.index.equals(comparison.index):
        raise ValueError("The portfolios do not match exactly in terms of stocks.")
    scaling = portfolio_spec["risk_aversion"] if portfolio_spec.get("risk_aversion") is not None else 1
    return np.abs(portfolio_weights_df * scaling - comparison).mean().item()


def rebalancing_schedule(trading_dates, rebalancing_frequency):
    """Rebalancing dates of a backtest (ref:1166-1176): they depend on the date list only, which is
    what lets `backtest_portfolio` solve all windows in one batch before replaying the P&L."""
    out, last = [], None
    for ts in trading_dates:
        if last is None or rebalancing_frequency == "daily":
            hit = True
        elif rebalancing_frequency == "weekly":
            hit = ts.weekday() == 2 or (ts - last).days > 7
        elif rebalancing_frequency == "monthly":
            hit = ts.month != last.month
        else:
            logger.error("Unknown rebalancing frequency.")
            raise ValueError("Unknown rebalancing frequency.")
        if hit:
            out.append(ts)
            last = ts
    return out


class Portfolio:
    """Daily P&L, weight drift, rebalancing, turnover and weight metrics (ref:1106-1219)."""

    def __init__(self, ts_start_date, portfolio_spec, precomputed_weights=None):
        self.ts_start_date = ts_start_date
        self.portfolio_spec = portfolio_spec
        self.portfolio_simple_returns_series = pd.Series(dtype="float64", name=portfolio_spec["display_name"])
        self.portfolio_turnover_series = pd.Series(dtype="float64", name=portfolio_spec["display_name"])
        self.portfolio_weights_metrics_df = pd.DataFrame(dtype="float64")
        self.last_rebalance_date_ts = None
        self._precomputed = precomputed_weights or {}

    def get_portfolio_simple_returns(self):
        return self.portfolio_simple_returns_series

    def get_portfolio_turnover(self):
        return self.portfolio_turnover_series

    def get_portfolio_weights_metrics(self):
        return self.portfolio_weights_metrics_df

    def _mark_to_market(self, trading_date_ts, market_data):
        w = self.portfolio_weights_df["Weight"]
        returns = market_data["stock_simple_returns_df"].loc[trading_date_ts].reindex(self.portfolio_weights_df.index)
        rf_annual = market_data["risk_free_rate_df"].asof(trading_date_ts).iloc[0]
        rf_daily = (rf_annual + 1) ** (1 / 252) - 1
        cash = 1 - w.sum()
        self.portfolio_simple_returns_series[trading_date_ts] = (returns * w).sum() + cash * rf_daily   # ref:1137-1145
        cash_after = cash * (1 + rf_daily)
        drifted = w * (1 + returns)
        total = drifted.sum() + cash_after
        self.portfolio_weights_df["Weight"] = drifted / total                                          # ref:1152-1159
        if abs((self.portfolio_weights_df["Weight"].values.sum() + cash_after / total) - 1) > 1e-5:
            logger.error("Weights do not sum to 1.")
            raise ValueError("Weights do not sum to 1.")

    def _is_rebalance_day(self, trading_date_ts):
        freq = self.portfolio_spec["rebalancing_frequency"]
        if self.last_rebalance_date_ts is None or freq == "daily":
            return True
        if freq == "weekly":
            return trading_date_ts.weekday() == 2 or (trading_date_ts - self.last_rebalance_date_ts).days > 7
        if freq == "monthly":
            return trading_date_ts.month != self.last_rebalance_date_ts.month
        logger.error("Unknown rebalancing frequency.")
        raise ValueError("Unknown rebalancing frequency.")

    def update_portfolio(self, trading_date_ts, market_data):
        if self.ts_start_date != trading_date_ts:
            self._mark_to_market(trading_date_ts, market_data)
        if not self._is_rebalance_day(trading_date_ts):
            return
        first = self.last_rebalance_date_ts is None
        before = None if first else self.portfolio_weights_df.copy()
        if trading_date_ts in self._precomputed:
            self.portfolio_weights_df = self._precomputed[trading_date_ts].copy()
        else:
            self.portfolio_weights_df = calculate_portfolio_weights(trading_date_ts, self.portfolio_spec, market_data)
        distance = calculate_average_distance_to_comparison_portfolio(self.portfolio_weights_df, self.portfolio_spec,
                                                                      trading_date_ts, market_data, "vw")
        w = self.portfolio_weights_df["Weight"]
        row = pd.DataFrame({"max_long": [w[w > 0].max()], "max_short": [w[w < 0].min()],
                            "avg_long": [w[w > 0].mean()], "avg_short": [w[w < 0].mean()],
                            "average_distance_to_comparison_portfolio": [distance]}, index=[trading_date_ts])
        self.portfolio_weights_metrics_df = pd.concat([self.portfolio_weights_metrics_df, row])
        if not first:
            turnover = compute_portfolio_turnover(before, self.portfolio_weights_df)
            self.portfolio_turnover_series[trading_date_ts] = turnover
            self.portfolio_simple_returns_series[trading_date_ts] -= self.portfolio_spec["turnover_cost"] / 10000 * turnover
        logger.info(f"Portfolio size {trading_date_ts}: {len(self.portfolio_weights_df.index)}")
        self.last_rebalance_date_ts = trading_date_ts


def _rf_asof(risk_free_rate_df, trading_dates):
    """`risk_free_rate_df.asof(ts).iloc[0]` for every date (ref:1139): the last row at or before ts that has no
    NaN (DataFrame.asof skips NaN rows)."""
    rf = risk_free_rate_df.sort_index()
    vals = rf.to_numpy(dtype=np.float64).reshape(len(rf), -1)
    ok = ~np.isnan(vals).any(axis=1)
    idx_ns = rf.index.values.astype("datetime64[ns]").astype(np.int64)[ok]
    first = vals[ok][:, 0]
    d = np.array([pd.Timestamp(ts).value for ts in trading_dates], dtype=np.int64)
    pos = np.searchsorted(idx_ns, d, side="right") - 1
    return np.where(pos >= 0, first[np.maximum(pos, 0)], np.nan)


def _replay_backtest(trading_dates, rebalance_dates, weights, cols, caps, tickers, portfolio_spec, market_data):
    """The daily loop of ref:1127-1219 (mark to market, drift, rebalance, turnover ref:1054-1075, weight metrics,
    distance to the value-weighted portfolio ref:1077-1104) over plain arrays: the weights of every rebalancing
    date are already in hand, so a day costs a few k-vector operations instead of a dozen DataFrame slices.
    Same operations in the same order as `Portfolio.update_portfolio`, NaN-skipping sums where pandas skips."""
    name = portfolio_spec["display_name"]
    K = len(tickers)
    sr = market_data["stock_simple_returns_df"]
    missing = [ts for ts in trading_dates[1:] if ts not in sr.index]
    if missing:
        raise KeyError(missing[0])                                        # ref:1134 `.loc[trading_date_ts]`
    S = sr.reindex(index=pd.DatetimeIndex(trading_dates), columns=tickers).to_numpy(dtype=np.float64)
    rf_daily_all = (_rf_asof(market_data["risk_free_rate_df"], trading_dates) + 1) ** (1 / 252) - 1     # ref:1140
    scaling = portfolio_spec["risk_aversion"] if portfolio_spec.get("risk_aversion") is not None else 1
    cost = portfolio_spec["turnover_cost"]
    reb_index = {ts: j for j, ts in enumerate(rebalance_dates)}
    n_days = len(trading_dates)
    returns = np.full(n_days, np.nan)
    turnover = np.full(len(rebalance_dates), np.nan)
    metrics = np.full((len(rebalance_dates), 5), np.nan)
    w = held = None
    with np.errstate(invalid="ignore", divide="ignore"):
        for di, ts in enumerate(trading_dates):
            if di > 0:
                r, rf_daily = S[di, held], rf_daily_all[di]
                cash = 1 - np.nansum(w)
                returns[di] = np.nansum(r * w) + cash * rf_daily                    # ref:1137-1145
                cash_after = cash * (1 + rf_daily)
                drifted = w * (1 + r)
                total = np.nansum(drifted) + cash_after
                w = drifted / total                                                 # ref:1152-1159
                if abs((w.sum() + cash_after / total) - 1) > 1e-5:
                    logger.error("Weights do not sum to 1.")
                    raise ValueError("Weights do not sum to 1.")
            j = reb_index.get(ts)
            if j is None:
                continue
            before_w, before_held = w, held
            w, held = weights[j].astype(np.float64, copy=True), cols[j]
            vw = caps[j] / np.nansum(caps[j])                                       # ref:692-695
            pos_w, neg_w = w[w > 0], w[w < 0]
            metrics[j] = (pos_w.max() if pos_w.size else np.nan, neg_w.min() if neg_w.size else np.nan,
                          pos_w.mean() if pos_w.size else np.nan, neg_w.mean() if neg_w.size else np.nan,
                          np.nanmean(np.abs(w * scaling - vw)))                      # ref:1100-1104
            if before_w is not None:
                dense_b, dense_a = np.zeros(K), np.zeros(K)                         # outer merge + fillna(0), ref:1062-1066
                dense_b[before_held] = np.nan_to_num(before_w, nan=0.0)
                dense_a[held] = np.nan_to_num(w, nan=0.0)
                union = np.union1d(before_held, held)
                traded = np.abs(dense_b[union] - dense_a[union]).sum()
                turnover[j] = (traded + abs(np.nansum(before_w) - np.nansum(w))) / 2    # ref:1068-1075
                returns[di] -= cost / 10000 * turnover[j]                           # ref:1212
    days = pd.DatetimeIndex(trading_dates)
    reb = pd.DatetimeIndex(rebalance_dates)
    return {"portfolio_simple_returns_series": pd.Series(returns[1:], index=days[1:], name=name, dtype="float64"),
            "portfolio_turnover_series": pd.Series(turnover[1:], index=reb[1:], name=name, dtype="float64"),
            "portfolio_weights_metrics_df": pd.DataFrame(
                metrics, index=reb, dtype="float64",
                columns=["max_long", "max_short", "avg_long", "avg_short", "average_distance_to_comparison_portfolio"])}


def backtest_portfolio(portfolio_spec, ts_start_date, ts_end_date, market_data):
    """Same inputs and outputs as ref:1221-1238.  All posterior solves of the backtest happen in ONE device
    batch, and the daily replay runs over arrays (`_replay_backtest`); `Portfolio.update_portfolio` remains the
    day-at-a-time form of the same loop."""
    trading_dates = [pd.Timestamp(ts) for ts in market_data["stock_prices_df"].index]
    trading_dates = [ts for ts in trading_dates if ts_start_date <= ts <= ts_end_date]
    rebalance_dates = rebalancing_schedule(trading_dates, portfolio_spec["rebalancing_frequency"])
    _prefetch_siblings(portfolio_spec, rebalance_dates, market_data)
    weights, labels, cols, caps = _weights_for_dates(rebalance_dates, portfolio_spec, market_data)
    tickers = batch.panels_for(market_data, portfolio_spec["rolling_window_frequency"]).tickers
    return _replay_backtest(trading_dates, rebalance_dates, weights, cols, caps, tickers, portfolio_spec, market_data)


def backtest_portfolios(portfolio_specs_dict, ts_start_date, ts_end_date, market_data):
    """`backtest_portfolio` for a whole spec grid ({name: spec}, as `create_portfolio_specs` returns it): the
    conjugate specs that share their windows go to the device in ONE batch (`calculate_weights_for_specs`), then
    every spec is replayed.  Specs outside the scope of this build (shrinkage, Black-Litterman: pypfopt) are
    skipped with a warning unless the caller supplied them (`register_strategy`)."""
    trading_dates = [pd.Timestamp(ts) for ts in market_data["stock_prices_df"].index]
    trading_dates = [ts for ts in trading_dates if ts_start_date <= ts <= ts_end_date]
    groups = {}
    for name, sp in portfolio_specs_dict.items():
        if sp["weighting_strategy"] in _CONJUGATE:
            groups.setdefault(_batch_family(sp), []).append(sp)
    for family, specs in groups.items():
        if len(specs) > 1:
            calculate_weights_for_specs(rebalancing_schedule(trading_dates, specs[0]["rebalancing_frequency"]), specs, market_data)
    out = {}
    for name, sp in portfolio_specs_dict.items():
        if sp["weighting_strategy"] in _OUT_OF_SCOPE and sp["weighting_strategy"] not in _EXTERNAL_STRATEGIES:
            logger.warning(f"{name}: {sp['weighting_strategy']} is outside the scope of this build (pypfopt); skipped.")
            continue
        out[name] = backtest_portfolio(sp, ts_start_date, ts_end_date, market_data)
    return out
