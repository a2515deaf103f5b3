"""Batch-native host packing: all rebalancing dates of a backtest are turned into
ONE device batch in the panel + row-index + column-index layout of `include/tangency_posterior.h`,
without touching pandas per date.

The reference re-slices DataFrames for every date (`/root/reference/src/portfolio_calculations.py`
ref:611-658 universe selection with a Python loop over tickers, ref:136-161 window + resample, ref:31-62
log-returns twice per window, ref:299-314 intraday filter, ref:90-114 MCM window).  Here every frame is
converted once per `market_data` into numpy panels (`MarketPanels`); a date then costs a few
`searchsorted` calls and small index arithmetic, and the O(n k) values never leave the shared panels:

* daily panel      PRICES (daily, or the weekly / monthly `resample().last()` bins plus one price row per date
                   for the running bin, ref:149-156) and the row pairs (numerator, denominator) of every
                   log-return row `L[i] = log(P[num_i] / P[den_i])`: the device forms the return panel
                   (`tp_inputs_t.ret_num`); the host only keeps WHICH returns are NaN
* intraday panel   bar prices and the pairs (i, i-1), likewise
* per window       row indices into the return panels, the k column indices in market-cap order, the per-row
                   risk-free adjustment (ref:40-57), prior weights w0 and strength n0

`tests/test_host_batch_packing.py` holds this packer to the frame-based one
(`portfolio_calculations._pack_window`, a line-by-line mirror of the reference's slicing) on every date.
"""
from __future__ import annotations

import logging

import numpy as np
import pandas as pd

logger = logging.getLogger(__name__)

_NS_PER_DAY = 86_400_000_000_000
_RESAMPLE_RULE = {"weekly": "W", "monthly": "ME"}
_CALENDAR_DAYS = {"daily": 1, "weekly": 7, "monthly": 31}
_TRADING_DAYS = {"daily": 1, "weekly": 5, "monthly": 22}


def _ns(index) -> np.ndarray:
    return pd.DatetimeIndex(index).values.astype("datetime64[ns]").astype(np.int64)


def _resample_last(df, frequency):
    try:
        return df.resample(_RESAMPLE_RULE[frequency]).last()
    except ValueError:  # pandas < 2.2
        return df.resample("M" if frequency == "monthly" else "W").last()


def _return_is_nan(num, den):
    """Where log(num / den) would be NaN (a missing price, or a ratio that is negative or 0/0): the host only
    needs the mask (ref:60 dropna, ref:646 eligibility); the logarithm itself is taken on the device."""
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = num / den
    return np.isnan(ratio) | (ratio < 0)


class MarketPanels:
    """numpy views of `market_data`, built once per (market_data, rolling_window_frequency)."""

    def __init__(self, market_data, frequency):
        if frequency not in _CALENDAR_DAYS:
            raise RuntimeError("Unknown rolling window frequency.")
        self.frequency = frequency
        prices = market_data["stock_prices_df"].sort_index()
        self.tickers = list(prices.columns)
        self.col_of = {t: i for i, t in enumerate(self.tickers)}
        self.dates = prices.index
        self.date_ns = _ns(prices.index)
        self.P = prices.to_numpy(dtype=np.float64)
        D, K = self.P.shape
        # length of the run of non-NaN prices ending at each row (ref:646 `.tail(n).notna().all()`)
        # = row index minus the row of the last NaN at or above it (a running maximum down the columns: no Python loop)
        ok = ~np.isnan(self.P)
        rows = np.arange(D, dtype=np.int32)[:, None]
        last_nan = np.maximum.accumulate(np.where(ok, np.int32(-1), rows), axis=0) if D else np.zeros((0, K), np.int32)
        self.valid_run = (rows - last_nan).astype(np.int32)

        caps = market_data["stock_market_caps_df"]
        self.caps_has = np.array([t in caps.columns for t in self.tickers])
        self.caps_ns = _ns(caps.index)
        self.caps = caps.reindex(columns=self.tickers).to_numpy(dtype=np.float64)

        hf = market_data["stock_intraday_prices_df"].sort_index()
        self.hf_has = np.array([t in hf.columns for t in self.tickers])
        self.hf_ns = _ns(hf.index)
        Hp = hf.reindex(columns=self.tickers).to_numpy(dtype=np.float64)
        self.Hp = np.ascontiguousarray(Hp)                     # bar prices; return i = bar i-1 -> bar i (device)
        self.H_nan = np.ones(Hp.shape, dtype=bool)
        if len(Hp) > 1:
            self.H_nan[1:] = _return_is_nan(Hp[1:], Hp[:-1])
        self.hf_notna_cum = np.concatenate([np.zeros((1, K), np.int32), np.cumsum(~np.isnan(Hp), axis=0, dtype=np.int32)])

        rf = market_data["risk_free_rate_df"]
        self.rf_ns = _ns(rf.index)
        self.rf = rf.to_numpy(dtype=np.float64).reshape(len(rf), -1)[:, 0]

        if frequency == "daily":
            self.base = self.P                                 # price rows the return rows are formed from
            self.label_ns = self.date_ns
        else:
            R = _resample_last(prices, frequency)              # complete bins are the same for every date
            self.R = np.ascontiguousarray(R.to_numpy(dtype=np.float64))
            self.base = self.R
            self.label_ns = _ns(R.index)
            # bin of each trading date: first label >= date (labels are bin ends)
            self.bin_of = np.searchsorted(self.label_ns, self.date_ns, side="left")
        self.L_nan = np.ones(self.base.shape, dtype=bool)      # return row i: base row i-1 -> base row i
        if len(self.base) > 1:
            self.L_nan[1:] = _return_is_nan(self.base[1:], self.base[:-1])
        # Running counts, so that "is any return of rows a..b of these columns NaN" is O(k) per date instead of a
        # gather of the whole window (the gathers were half of pack_windows' time): per column for the daily / binned
        # returns, over all columns for the (much longer) intraday panel.
        self.L_nan_cum = np.concatenate([np.zeros((1, K), np.int32), np.cumsum(self.L_nan, axis=0, dtype=np.int32)])
        self.H_rownan_cum = np.concatenate([[0], np.cumsum(self.H_nan.any(axis=1), dtype=np.int64)])
        # whole days between consecutive labels (`.dt.days` of the label differences, ref:40) and the risk-free value
        # each label takes by forward fill (ref:54), once per panel
        self.label_gap_days = np.diff(self.label_ns) // _NS_PER_DAY
        ridx = np.searchsorted(self.rf_ns, self.label_ns, side="right") - 1
        self.rf_at_label = np.where(ridx >= 0, self.rf[np.maximum(ridx, 0)], np.nan) if len(self.rf) else np.full(len(self.label_ns), np.nan)
        self.ticker_arr = np.array(self.tickers, dtype=object)
        self._mcm = {}
        self.weights_cache = {}        # (spec key, dates) -> device results, filled by cross-spec batches

    # -- market-condition metric (ref:90-114, 247-267) --------------------------------------------
    def mcm(self, market_data, key):
        src = market_data[key]
        hit = self._mcm.get(key)
        if hit is None or hit["frame"] is not src:      # a replaced VIX / EPU frame must not reuse the old arrays
            frame = src.sort_index()
            vals = frame.to_numpy(dtype=np.float64).reshape(len(frame), -1)[:, 0]
            entry = dict(ns=_ns(frame.index), vals=vals)
            if self.frequency != "daily":
                R = _resample_last(frame, self.frequency)
                entry["rvals"] = R.to_numpy(dtype=np.float64).reshape(len(R), -1)[:, 0]
                entry["rlabel_ns"] = _ns(R.index)
            entry["frame"] = src
            self._mcm[key] = entry
        return self._mcm[key]


_PANEL_CACHE = {}
_PANEL_FRAMES = ("stock_prices_df", "stock_intraday_prices_df", "stock_market_caps_df", "risk_free_rate_df")


def panels_for(market_data, frequency) -> MarketPanels:
    """The numpy panels of `market_data`, cached per (frames, frequency).  The cache entry holds a reference to
    EVERY frame it was built from and is valid only while `market_data` still holds those very objects (an `id()`
    alone could be recycled by a new frame).  Frames must not be edited in place between calls - replace them, or
    call `clear_panel_cache()`."""
    frames = tuple(market_data[name] for name in _PANEL_FRAMES)
    key = tuple(id(f) for f in frames) + (frequency,)
    hit = _PANEL_CACHE.get(key)
    if hit is None or any(a is not b for a, b in zip(hit[0], frames)):
        if len(_PANEL_CACHE) > 8:
            _PANEL_CACHE.clear()
        hit = (frames, MarketPanels(market_data, frequency))
        _PANEL_CACHE[key] = hit
    return hit[1]


def clear_panel_cache():
    """Forget every cached panel (and the weights cached with them), e.g. after editing a frame in place."""
    _PANEL_CACHE.clear()


def _mean_gap_and_check(gaps):
    """`gaps`: whole days between consecutive labels of the window (MarketPanels.label_gap_days slice)."""
    mean_gap = gaps.mean()
    assert gaps.max() <= mean_gap + 4, "Unexpected large gap between return dates."
    return mean_gap


def select_universe(mp: MarketPanels, pos, size, window_days, rebal_frequency, members):
    """Column indices (into mp.tickers) of the `size` largest caps among eligible stocks, cap-descending
    (ref:611-658; the caller passes the REBALANCING frequency as in ref:960, Appendix B-Q7)."""
    if rebal_frequency not in _CALENDAR_DAYS:
        logger.error("Unknown rolling window frequency.")
        raise RuntimeError("Unknown rolling window frequency.")
    span = _CALENDAR_DAYS[rebal_frequency]
    d = mp.date_ns[pos]
    need = min(window_days, pos + 1)
    elig = members & mp.caps_has & mp.hf_has & (mp.valid_run[pos] >= need)
    a = np.searchsorted(mp.hf_ns, d - span * _NS_PER_DAY, side="left")     # label slice, both ends inclusive
    b = np.searchsorted(mp.hf_ns, d + _NS_PER_DAY, side="right")
    elig &= (mp.hf_notna_cum[b] - mp.hf_notna_cum[a]) > 0
    ci = np.searchsorted(mp.caps_ns, d)
    if ci >= len(mp.caps_ns) or mp.caps_ns[ci] != d:
        ts = mp.dates[pos]
        logger.error(f"The trading date {ts} does not exist in the market capitalizations data.")
        raise ValueError(f"The trading date {ts} does not exist in the market capitalizations data.")
    caps = mp.caps[ci]
    cand = np.flatnonzero(elig & ~np.isnan(caps))
    order = cand[np.argsort(-caps[cand], kind="stable")][:size]            # nlargest keeps the first of ties
    return order.astype(np.int32), caps[order]


def pack_universes(trading_dates, portfolio_spec, market_data, members_of=None):
    """Universe selection only (ref:953-988): (column indices [W x k], labels, market caps [W x k]) per date.
    What the passive strategies vw / ew need - the reference touches neither returns nor the risk-free rate for
    them (ref:990-997), so none of the estimator-only conditions (date gaps, NaN risk-free days, intraday bars of
    the window) may stop them."""
    freq = portfolio_spec["rolling_window_frequency"]
    N, k = portfolio_spec["rolling_window"], portfolio_spec["size"]
    mp = panels_for(market_data, freq)
    window_days = N * _TRADING_DAYS[freq]
    W, K = len(trading_dates), len(mp.tickers)
    col_idx = np.zeros((W, k), dtype=np.int32)
    caps_all = np.zeros((W, k), dtype=np.float64)
    all_members = np.ones(K, dtype=bool)
    for w, ts in enumerate(trading_dates):
        pos, members = _date_position_and_members(mp, ts, members_of, all_members)
        cols, caps = select_universe(mp, pos, k, window_days, portfolio_spec["rebalancing_frequency"], members)
        if len(cols) != k:
            raise ValueError(f"universe has {len(cols)} assets, portfolio_spec['size'] is {k}")
        col_idx[w], caps_all[w] = cols, caps
        lo_row = max(0, pos + 1 - window_days)                     # ref:986-988
        if (mp.valid_run[pos, cols] < pos + 1 - lo_row).any():
            logger.error("Found NA values in the filtered stock prices.")
            raise ValueError("The filtered stock prices contain NA values.")
    return col_idx, mp.ticker_arr[col_idx].tolist(), caps_all


def _date_position_and_members(mp, ts, members_of, all_members):
    d = pd.Timestamp(ts).value
    pos = int(np.searchsorted(mp.date_ns, d))
    if pos >= len(mp.date_ns) or mp.date_ns[pos] != d:
        raise ValueError(f"trading_date_ts {ts} must be the last date in the DataFrame.")
    members = all_members
    if members_of is not None:
        tick = members_of(ts)
        if tick is not None:
            ts_set = set(tick)
            members = np.array([t in ts_set for t in mp.tickers])
    return pos, members


def prior_inputs(trading_dates, portfolio_spec, market_data, caps):
    """(w0 [W x k], n0 [W]) of a conjugate spec for dates whose universes (market caps `caps`, cap-descending) are
    already selected: what differs between the conjugate specs of one grid (ref:361-380 vw / ew, ref:247-267 VIX /
    EPU and mcm_scaling), so that several specs can share ONE packed batch."""
    strategy = portfolio_spec["weighting_strategy"]
    freq = portfolio_spec["rolling_window_frequency"]
    N, k = portfolio_spec["rolling_window"], portfolio_spec["size"]
    mp = panels_for(market_data, freq)
    mcm = mp.mcm(market_data, "vix_prices_df" if "_vix_" in strategy else "epu_prices_df")
    caps = np.asarray(caps, dtype=np.float64)
    if "vw" in strategy:
        w0 = caps / caps.sum(axis=1, keepdims=True)                  # ref:692-695 (already cap-descending)
    elif "ew" in strategy:
        w0 = np.full(caps.shape, 1 / k)                              # ref:670-672
    else:
        logger.error("Unknown conjugate portfolio prior weights.")
        raise ValueError("Unknown conjugate portfolio prior weights.")
    n0 = np.array([_prior_strength(mp, mcm, pd.Timestamp(ts).value, ts, N, portfolio_spec["mcm_scaling"])
                   for ts in trading_dates], dtype=np.float64)
    return w0, n0


def pack_windows(trading_dates, portfolio_spec, market_data, members_of=None, return_caps=False):
    """All rebalancing dates of one spec -> keyword arguments of `_native.posterior_batch` plus the
    per-window ticker labels (and, with `return_caps`, the market caps [W x k] of the selected assets in the
    same order: the value-weighted comparison portfolio of ref:1077-1104).  Raises the reference's exceptions
    for the reference's conditions."""
    strategy = portfolio_spec["weighting_strategy"]
    freq = portfolio_spec["rolling_window_frequency"]
    N, k = portfolio_spec["rolling_window"], portfolio_spec["size"]
    conj = strategy.startswith("conjugate")
    mp = panels_for(market_data, freq)
    window_days = N * _TRADING_DAYS[freq]
    W = len(trading_dates)
    K = len(mp.tickers)
    n_r_max = N - 1

    col_idx = np.zeros((W, k), dtype=np.int32)
    caps_all = np.zeros((W, k), dtype=np.float64)
    row_idx = np.zeros((W, n_r_max), dtype=np.int32)
    n_rows = np.zeros(W, dtype=np.int32)
    rf_adj = np.zeros((W, n_r_max), dtype=np.float64)
    tails = []                                           # extra price rows (date prices of running bins) + their denominators
    w0 = np.zeros((W, k)) if conj else None
    n0 = np.zeros(W) if conj else None
    hf_rows, hf_count = [], np.zeros(W, dtype=np.int32)
    base_rows = mp.base.shape[0]
    mcm = mp.mcm(market_data, "vix_prices_df" if "_vix_" in strategy else "epu_prices_df") if conj else None
    all_members = np.ones(K, dtype=bool)

    for w, ts in enumerate(trading_dates):
        d = pd.Timestamp(ts).value
        pos, members = _date_position_and_members(mp, ts, members_of, all_members)
        cols, caps = select_universe(mp, pos, k, window_days, portfolio_spec["rebalancing_frequency"], members)
        if len(cols) != k:
            raise ValueError(f"universe has {len(cols)} assets, portfolio_spec['size'] is {k}")
        col_idx[w] = cols
        caps_all[w] = caps
        # ref:986-988: the filtered prices of the window must be complete (the run of non-NaN prices that ends at
        # the date covers the window: O(k), not a gather of the window)
        lo_row = max(0, pos + 1 - window_days)
        if (mp.valid_run[pos, cols] < pos + 1 - lo_row).any():
            logger.error("Found NA values in the filtered stock prices.")
            raise ValueError("The filtered stock prices contain NA values.")

        # ---- daily window rows (ref:136-161, 31-62)
        if freq == "daily":
            first_price = max(0, pos - N + 1)
            rows = np.arange(first_price + 1, pos + 1, dtype=np.int64)        # returns of prices first_price..pos
            lab0, lab1 = first_price, pos                                      # labels lab0..lab1 of mp.label_ns
        else:
            b = int(mp.bin_of[pos])
            first_bin = max(0, b - N + 1)
            body = np.arange(first_bin + 1, b, dtype=np.int64)                 # returns between complete bins
            # running bin: its return is P(date) over the last complete bin (a price row of its own for the device)
            tail_nan = _return_is_nan(mp.P[pos], mp.R[b - 1]) if b > first_bin else None
            rows = body
            if tail_nan is not None:
                tails.append((pos, b - 1))
                rows = np.concatenate([body, [base_rows + len(tails) - 1]])
            lab0, lab1 = first_bin, b
        if lab1 > lab0:
            mean_gap = _mean_gap_and_check(mp.label_gap_days[lab0:lab1])
            rfv = mp.rf_at_label[lab0 + 1:lab1 + 1]                            # ffill on labels (ref:54)
            adj = (1 + rfv) ** (mean_gap / 365) - 1                            # ref:48
        else:
            raise AssertionError("Unexpected large gap between return dates.")
        # dropna (ref:60): a NaN risk-free value or a NaN return of a selected asset drops the row
        if freq == "daily":
            if len(rows) and (mp.L_nan_cum[pos + 1, cols] != mp.L_nan_cum[first_price + 1, cols]).any():
                vals_nan = mp.L_nan[rows][:, cols].any(axis=1)
            else:
                vals_nan = np.zeros(len(rows), dtype=bool)
        else:
            block = mp.L_nan[body][:, cols]
            if tail_nan is not None:
                block = np.vstack([block, tail_nan[cols][None, :]])
            vals_nan = block.any(axis=1)
        keep = ~vals_nan & ~np.isnan(adj)
        rows, adj = rows[keep], adj[keep]
        n_rows[w] = len(rows)
        row_idx[w, :len(rows)] = rows
        rf_adj[w, :len(rows)] = adj

        if conj:
            # ---- intraday rows (ref:299-314): bars in (date + 1d - Delta, date + 1d]
            span = _CALENDAR_DAYS[freq]
            a = int(np.searchsorted(mp.hf_ns, d - span * _NS_PER_DAY + _NS_PER_DAY, side="right"))
            # ref:972-974 truncates the intraday frame at 23:59:59 of the trading date before ref:311-312
            end_of_day = pd.Timestamp(ts).replace(hour=23, minute=59, second=59).value
            e = int(np.searchsorted(mp.hf_ns, min(d + _NS_PER_DAY, end_of_day), side="right"))
            cand = np.arange(a + 1, e, dtype=np.int64)                          # the first bar's return is NaN (shift)
            if len(cand) and mp.H_rownan_cum[e] != mp.H_rownan_cum[a + 1]:   # some bar of the span has a NaN somewhere
                cand = cand[~mp.H_nan[cand][:, cols].any(axis=1)]
            hf_rows.append(cand)
            hf_count[w] = len(cand)
            # ---- prior weights and strength
            if "vw" in strategy:
                w0[w] = caps / caps.sum()                                       # ref:692-695 (already cap-descending)
            else:
                w0[w] = 1 / k                                                   # ref:670-672
            n0[w] = _prior_strength(mp, mcm, d, ts, N, portfolio_spec["mcm_scaling"])

    labels = mp.ticker_arr[col_idx].tolist()
    if (n_rows < 1).any():
        raise ValueError("a rolling window has no usable return rows")
    # Price panel + the (numerator, denominator) rows of every return row; the device takes the logarithms.
    # Return row i < base_rows is base row i over base row i-1 (row 0: itself, a zero row nobody selects); the
    # running-bin rows follow, each the date's price row (appended below the bins) over its last complete bin.
    nb = base_rows
    ret_num = np.arange(nb + len(tails), dtype=np.int32)
    ret_den = np.concatenate([[0], np.arange(nb - 1), [t[1] for t in tails]]).astype(np.int32)
    panel = mp.base if not tails else np.vstack([mp.base, mp.P[[t[0] for t in tails]]])
    kw = dict(panel=panel, ret_pairs=(ret_num, ret_den), start=None, n_r=int(n_rows.max()),
              row_idx=np.ascontiguousarray(row_idx[:, :int(n_rows.max())]), n_rows=n_rows, col_idx=col_idx,
              rf_adj=np.ascontiguousarray(rf_adj[:, :int(n_rows.max())]))
    if conj:
        if (hf_count < 2).any():
            raise ValueError("conjugate prior needs at least two intraday returns in the last period")
        m = int(hf_count.max())
        hidx = np.zeros((W, m), dtype=np.int32)
        for w, r in enumerate(hf_rows):
            hidx[w, :len(r)] = r
        nh = mp.Hp.shape[0]
        kw.update(hf_panel=mp.Hp, hf_ret_pairs=(np.arange(nh, dtype=np.int32), np.maximum(np.arange(nh) - 1, 0).astype(np.int32)),
                  hf_start=None, hf_row_idx=hidx, hf_count=hf_count, m=m, w0=w0, n0=n0)
    if return_caps:
        return kw, labels, caps_all
    return kw, labels


def _prior_strength(mp, mcm, d, ts, N, scaling):
    """n0 = N * max(cur/avg, avg/cur) * mcm_scaling (ref:90-114, 247-267) from the cached MCM arrays."""
    pos = int(np.searchsorted(mcm["ns"], d, side="right")) - 1
    if pos < 0 or mcm["ns"][pos] != d:
        logger.error(f"trading_date_ts {ts} is not the last date in the DataFrame.")
        raise ValueError(f"trading_date_ts {ts} must be the last date in the DataFrame.")
    cur = mcm["vals"][pos]
    if mp.frequency == "daily":
        window = mcm["vals"][max(0, pos - N + 1):pos + 1]
    else:
        b = int(np.searchsorted(mcm["rlabel_ns"], d, side="left"))
        # running bin: last valid observation of the bin up to the date (resample().last() skips NaN)
        lo = int(np.searchsorted(mcm["ns"], mcm["rlabel_ns"][b - 1], side="right")) if b >= 1 else 0
        seg = mcm["vals"][lo:pos + 1]
        seg = seg[~np.isnan(seg)]
        tail = seg[-1] if len(seg) else np.nan
        window = np.concatenate([mcm["rvals"][max(0, b - N + 1):b], [tail]])
    window = window[~np.isnan(window)]                                        # DataFrame.mean skips NaN
    avg = window.sum() / len(window)
    frac = cur / avg if cur > avg else avg / cur
    return N * frac * scaling
