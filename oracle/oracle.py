"""CPU oracle (numpy) for the rolling-window Bayesian tangency-portfolio posterior.

TEST INFRASTRUCTURE ONLY.  This module restates, on plain numpy arrays, the arithmetic of the
reference's hot path (`/root/reference/src/portfolio_calculations.py`, cited per function as
`ref:LINE`).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it; the product package (`incorporating_different_sources_amd/`) never does.

Parity pin: `tests/test_oracle_golden.py` checks every function here against golden vectors produced by
running the unmodified reference in the build container (`oracle/gen_golden.py` -> `tests/golden/*.npz`).

Conventions: one window = `X` (n_r x k excess log-returns, ref:31-62), `Y` (m x k intraday log-returns,
ref:314), `w0` (k prior weights, ref:361-380), scalars `n0` (ref:247-267), `N` = spec["rolling_window"],
`k` = spec["size"], `gamma` = spec["risk_aversion"].  Everything is IEEE fp64.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------------------------
# status codes shared with include/tangency_posterior.h
STATUS_OK = 0
STATUS_NOT_PD = 1      # a non-positive pivot in the Cholesky of S1 / J  (device only; LU has no such check)
STATUS_NONFINITE = 2   # NaN/Inf in the weights (ref:492-494 raises ValueError on NaN w1)
STATUS_BAD_DENOM = 3   # n1 - q1 <= 0 (ref:573 has no guard, Appendix B-Q9)


# ----------------------------------------------------------------------------------------------
# window preparation (ref:31-62, 136-161)
def excess_log_returns_from_prices(prices: np.ndarray, rf_adj: np.ndarray | None = None) -> np.ndarray:
    """ref:37  log(P_t / P_{t-1}); ref:57 subtract the per-period risk-free rate; ref:60 drop first row.

    `rf_adj` is the already frequency-adjusted per-row rate `(1+rf)^(dbar/365) - 1` (ref:48) aligned to
    the return rows (length n-1); None means rf = 0.
    """
    lr = np.log(prices[1:] / prices[:-1])
    if rf_adj is not None:
        lr = lr - np.asarray(rf_adj, dtype=np.float64)[:, None]
    return lr


def rf_adjusted(rf_annual: np.ndarray, mean_gap_days: float) -> np.ndarray:
    """ref:48  (1 + rf)^(dbar/365) - 1 with dbar the mean calendar-day gap of the window's price dates."""
    return (1.0 + np.asarray(rf_annual, dtype=np.float64)) ** (mean_gap_days / 365.0) - 1.0


# ----------------------------------------------------------------------------------------------
# canonical statistics (ref:163-245)
def canonical_statistics_T(X: np.ndarray) -> np.ndarray:
    """ref:180-182  T = X' X  (k x k)."""
    return X.T @ X


def canonical_statistics_t(X: np.ndarray) -> np.ndarray:
    """ref:222  t = sum_i x_i  (k)."""
    return X.sum(axis=0)


# ----------------------------------------------------------------------------------------------
# conjugate prior hyper-parameters (ref:90-114, 247-333, 382-430)
def conjugate_prior_n(mcm_window: np.ndarray, N: int, mcm_scaling: float = 1.0) -> float:
    """ref:112 avg = mean of the last N MCM observations (incl. today); ref:257 cur = today's value;
    ref:260-265 n0 = N * max(cur/avg, avg/cur) * mcm_scaling."""
    mcm_window = np.asarray(mcm_window, dtype=np.float64)
    avg = mcm_window[-N:].mean()
    cur = mcm_window[-1]
    frac = cur / avg if cur > avg else avg / cur
    return float(N * frac * mcm_scaling)


def conjugate_posterior_n(n0: float, N: int) -> float:
    """ref:282  n1 = n0 + N."""
    return n0 + N


def hf_scatter(Y: np.ndarray) -> np.ndarray:
    """ref:317-318  DataFrame.cov() (centered, ddof=1) times len(Y):  m/(m-1) * (Y-Ybar)'(Y-Ybar)."""
    m = Y.shape[0]
    Yc = Y - Y.mean(axis=0)
    return (Yc.T @ Yc) / (m - 1) * m


def conjugate_prior_S(Y: np.ndarray, n0: float) -> np.ndarray:
    """ref:333  S0 = n0 * (cov(Y) * m)."""
    return n0 * hf_scatter(Y)


def portfolio_variance(w: np.ndarray, S: np.ndarray) -> float:
    """ref:78  w' S w."""
    return float(w @ (S @ w))


def conjugate_c(n0: float, k: int, q0: float) -> float:
    """ref:415-418  c = 2 n0 / (a + sqrt(a^2 + 4 n0 q0)),  a = n0 + k + 2,  q0 = w0' S0 w0."""
    a = n0 + k + 2
    return (2 * n0) / (a + (a ** 2 + 4 * n0 * q0) ** (1 / 2))


def conjugate_posterior_S(S0: np.ndarray, T: np.ndarray) -> np.ndarray:
    """ref:358  S1 = S0 + T."""
    return S0 + T


def conjugate_posterior_w(S1: np.ndarray, S0: np.ndarray, w0: np.ndarray, t: np.ndarray, c: float) -> np.ndarray:
    """ref:485-489  w1 = inv(S1) . (c * S0 w0 + t)   (explicit LU inverse, Appendix B-Q8)."""
    return np.linalg.inv(S1) @ (c * (S0 @ w0) + t)


def mean_conjugate_posterior_nu(n1: float, k: int, w1: np.ndarray, S1: np.ndarray) -> np.ndarray:
    """ref:572-575  nu = (n1 + k + 2) * w1 / (n1 - w1' S1 w1)."""
    return (n1 + k + 2) * w1 / (n1 - portfolio_variance(w1, S1))


def conjugate_window(X, Y, w0, n0, N, k, gamma, return_aux=False):
    """ref:819-836 (calculate_conjugate_hf_mcm_portfolio) for one window: weights = nu / gamma."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    w0 = np.asarray(w0, dtype=np.float64)
    T = canonical_statistics_T(X)
    t = canonical_statistics_t(X)
    S0 = conjugate_prior_S(Y, n0)
    q0 = portfolio_variance(w0, S0)
    c = conjugate_c(n0, k, q0)
    S1 = conjugate_posterior_S(S0, T)
    w1 = conjugate_posterior_w(S1, S0, w0, t, c)
    n1 = conjugate_posterior_n(n0, N)
    q1 = portfolio_variance(w1, S1)
    nu = mean_conjugate_posterior_nu(n1, k, w1, S1)
    weights = 1 / gamma * nu
    if return_aux:
        return weights, dict(T=T, t=t, S0=S0, q0=q0, c=c, S1=S1, w1=w1, n1=n1, q1=q1, nu=nu)
    return weights


# ----------------------------------------------------------------------------------------------
# Jeffreys posterior (ref:580-608, 838-849)
def mean_jeffreys_posterior_nu(X: np.ndarray, N: int, rhs=None, shift=None) -> np.ndarray:
    """ref:600-606  J = T - (1/N) t t' ;  nu = inv(J) . t   (N = rolling_window, not n_r: Appendix B-Q1).
    `rhs` replaces t (the C-ABI's tp_batch_set_rhs); N = None drops the t t'/N term (TP_FLAG_NO_CENTER);
    `shift` = (d, e) adds d I + e 1 1' (tp_batch_set_shift)."""
    T = canonical_statistics_T(X)
    t = canonical_statistics_t(X)
    J = T - 1 / N * np.outer(t, t) if N is not None else T.copy()
    if shift is not None:
        J = J + shift[0] * np.eye(J.shape[0]) + shift[1] * np.ones_like(J)
    return np.linalg.inv(J) @ (t if rhs is None else rhs)


def jeffreys_window(X, N, gamma, return_aux=False, rhs=None, shift=None):
    """ref:838-849 (calculate_jeffreys_portfolio): weights = nu / gamma."""
    X = np.asarray(X, dtype=np.float64)
    nu = mean_jeffreys_posterior_nu(X, N, rhs, shift)
    weights = 1 / gamma * nu
    if return_aux:
        T = canonical_statistics_T(X)
        t = canonical_statistics_t(X)
        return weights, dict(T=T, t=t, J=T - 1 / N * np.outer(t, t), nu=nu)
    return weights


# ----------------------------------------------------------------------------------------------
# Jorion hyper-parameter (Bayes-Stein) portfolio (ref:851-895), SURVEY §8(f) row F3
def jorion_window(X, gamma):
    """ref:869-893 with the notation of the reference: N assets, T return rows."""
    X = np.asarray(X, dtype=np.float64)
    T, N = X.shape
    mu = X.mean(axis=0)                                             # ref:873
    V = np.cov(X, rowvar=False, ddof=1).reshape(N, N)               # ref:876
    Vbar = T / (T - N - 2) * V                                      # ref:879
    Vbi = np.linalg.inv(Vbar)                                       # ref:880
    one = np.ones(N)
    mu_g = (one @ Vbi @ mu) / (one @ Vbi @ one)                     # ref:882
    d = mu - mu_g * one
    lam = (N + 2) / (d @ Vbi @ d)                                   # ref:885
    v = (N + 2) / ((N + 2) + T * (d @ Vbi @ d))                     # ref:887
    V_PJ = (1 + 1 / (T + lam)) * Vbar + lam / (T * (T + 1 + lam)) * np.outer(one, one) / (one @ Vbi @ one)   # ref:888
    mu_PJ = (1 - v) * mu + v * mu_g * one                           # ref:889
    return 1 / gamma * (np.linalg.inv(V_PJ) @ mu_PJ)                # ref:891-893


# ----------------------------------------------------------------------------------------------
# Greyserman et al. hierarchical prior, Monte-Carlo mean over hyper-parameter draws (ref:897-938)
def greyserman_draws(count=1000):
    """The reference's draw sequence (ref:925-927): alternating numpy-global uniform(-1000, 1000) and
    scipy gamma(a=1, scale=10) variates.  Seed with numpy.random.seed for reproducible weights."""
    from scipy.stats import gamma as _gamma
    xi = np.empty(count)
    eta = np.empty(count)
    for i in range(count):
        xi[i] = np.random.uniform(-1000, 1000)
        eta[i] = _gamma.rvs(a=1, scale=10)
    return xi, eta


def greyserman_window(X, gamma, xi, eta):
    """ref:914-934 for one window of excess log-returns X [n x k] and given draws (xi_b, eta_b)."""
    X = np.asarray(X, dtype=np.float64)
    n, k = X.shape
    x_bar = X.mean(axis=0)[:, None]
    S = np.cov(X, rowvar=False, ddof=1).reshape(k, k)
    S_h = np.where(np.eye(k) == 1, 1, 0.5)
    one = np.ones((k, 1))
    kappa_h = round(0.1 * n)
    nu_h = k
    acc = np.zeros((k, 1))
    for xi_b, eta_b in zip(xi, eta):
        a_h = 1 / (n + kappa_h) * (n * x_bar + kappa_h * xi_b * one)
        D_h = ((n - 1) * S + eta_b * S_h + n * x_bar @ x_bar.T + kappa_h * xi_b ** 2 * one @ one.T
               - (n + kappa_h) * a_h @ a_h.T)
        acc += 1 / gamma * (nu_h + n + 1) * (1 - 1 / (nu_h + n - k)) * (np.linalg.inv(D_h) @ a_h)
    return (acc / len(xi))[:, 0]


def log_return_rows(prices, num, den):
    """ref:44 / ref:311 for arbitrary row pairs: log(P[num] / P[den]), then the packer's nan_to_num(nan=0)."""
    P = np.asarray(prices, dtype=np.float64)
    with np.errstate(all="ignore"):
        return np.nan_to_num(np.log(P[np.asarray(num)] / P[np.asarray(den)]), nan=0.0)


# ----------------------------------------------------------------------------------------------
# batched driver over the panel+offset layout of include/tangency_posterior.h (numpy loop; small cases)
def posterior_batch(strategy, k, N, gamma, panel, start, n_r, hf_panel=None, hf_start=None, m=None,
                    w0=None, n0=None, row_idx=None, n_rows=None, col_idx=None, rf_adj=None,
                    hf_row_idx=None, hf_count=None, rhs=None, center_rows=False, no_center=False, shift=None,
                    ret_pairs=None, hf_ret_pairs=None):
    """Loop `conjugate_window` / `jeffreys_window` over W windows described the way the C-ABI takes them.

    Returns (weights [W x k], status [W] int32, aux [W x 8] = n0, n1, c, q0, q1, denom, 0, 0).
    """
    W = len(start) if start is not None else len(row_idx)
    weights = np.empty((W, k))
    status = np.zeros(W, dtype=np.int32)
    aux = np.zeros((W, 8))
    if ret_pairs is not None:           # price front-end of the C-ABI (tp_inputs_t.ret_num / ret_den)
        panel = log_return_rows(panel, *ret_pairs)
    if hf_ret_pairs is not None:
        hf_panel = log_return_rows(hf_panel, *hf_ret_pairs)
    for w in range(W):
        nr = int(n_rows[w]) if n_rows is not None else n_r
        rows = (np.asarray(row_idx[w][:nr], dtype=np.int64) if row_idx is not None
                else np.arange(start[w], start[w] + nr))
        cols = np.asarray(col_idx[w], dtype=np.int64) if col_idx is not None else np.arange(k)
        X = panel[np.ix_(rows, cols)]
        if rf_adj is not None:
            X = X - np.asarray(rf_adj[w][:nr])[:, None]
        with np.errstate(all="ignore"):
            if strategy == "conjugate":
                mm = int(hf_count[w]) if hf_count is not None else m
                hrows = (np.asarray(hf_row_idx[w][:mm], dtype=np.int64) if hf_row_idx is not None
                         else np.arange(hf_start[w], hf_start[w] + mm))
                Y = hf_panel[np.ix_(hrows, cols)]
                wt, a = conjugate_window(X, Y, w0[w], float(n0[w]), N, k, gamma, return_aux=True)
                denom = a["n1"] - a["q1"]
                aux[w, :6] = (float(n0[w]), a["n1"], a["c"], a["q0"], a["q1"], denom)
                if not (denom > 0):
                    status[w] = STATUS_BAD_DENOM
            elif strategy == "jeffreys":
                wt = jeffreys_window(X, None if no_center else (X.shape[0] if center_rows else N), gamma,
                                     rhs=None if rhs is None else rhs[w], shift=None if shift is None else shift[w])
            else:
                raise ValueError("Unknown weights spec.")
        if not np.all(np.isfinite(wt)):
            status[w] = STATUS_NONFINITE
        weights[w] = wt
    return weights, status, aux


# ----------------------------------------------------------------------------------------------
# C restatement (oracle/tangency_oracle.c), loaded through ctypes
_LIB = None


def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/tangency_oracle.c -> oracle/liboracle_tangency.so (gcc, OpenMP)."""
    so = os.path.join(_HERE, "liboracle_tangency.so")
    src = os.path.join(_HERE, "tangency_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_tangency.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_tangency.so")
        if not os.path.exists(so):
            build_c_oracle()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_posterior_batch.restype = ctypes.c_int
        _LIB.oracle_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def c_num_threads() -> int:
    return int(_lib().oracle_num_threads())


def posterior_batch_c(strategy, k, N, gamma, panel, start, n_r, hf_panel=None, hf_start=None, m=0,
                      w0=None, n0=None, row_idx=None, n_rows=None, col_idx=None, rf_adj=None,
                      hf_row_idx=None, hf_count=None, threads=0, rhs=None, center_rows=False):
    """Same contract as `posterior_batch`, computed by the C restatement (OpenMP over windows)."""
    lib = _lib()
    f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    i64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int64)
    i32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
    panel = f64(panel); hf_panel = f64(hf_panel); w0 = f64(w0); n0 = f64(n0); rf_adj = f64(rf_adj)
    start = i64(start); hf_start = i64(hf_start)
    row_idx = i32(row_idx); n_rows = i32(n_rows); col_idx = i32(col_idx)
    hf_row_idx = i32(hf_row_idx); hf_count = i32(hf_count); rhs = f64(rhs)
    W = len(start) if start is not None else row_idx.shape[0]
    weights = np.empty((W, k)); status = np.zeros(W, dtype=np.int32); aux = np.zeros((W, 8))
    strat = {"conjugate": 0, "jeffreys": 1}[strategy]
    rc = lib.oracle_posterior_batch(
        ctypes.c_int(strat), ctypes.c_int(k), ctypes.c_int(N), ctypes.c_int(n_r), ctypes.c_int(m or 0),
        ctypes.c_double(gamma), ctypes.c_longlong(W),
        _p(panel, ctypes.c_double), ctypes.c_int(panel.shape[1]),
        _p(start, ctypes.c_longlong), _p(row_idx, ctypes.c_int), _p(n_rows, ctypes.c_int),
        _p(col_idx, ctypes.c_int), _p(rf_adj, ctypes.c_double),
        _p(hf_panel, ctypes.c_double), ctypes.c_int(hf_panel.shape[1] if hf_panel is not None else 0),
        _p(hf_start, ctypes.c_longlong), _p(hf_row_idx, ctypes.c_int), _p(hf_count, ctypes.c_int),
        _p(w0, ctypes.c_double), _p(n0, ctypes.c_double),
        _p(weights, ctypes.c_double), _p(status, ctypes.c_int), _p(aux, ctypes.c_double),
        ctypes.c_int(threads), _p(rhs, ctypes.c_double), ctypes.c_int(1 if center_rows else 0))
    if rc != 0:
        raise RuntimeError(f"oracle_posterior_batch failed rc={rc}")
    return weights, status, aux
