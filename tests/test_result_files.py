"""The on-disk result formats: the three files src/main.py:79-81 writes per
spec - `Series.to_csv(header=True)` twice, `DataFrame.to_csv(header=True)` once - and the cache-load call that reads them
back (src/main.py:56-66).  The golden holds the TEXT the unmodified reference's results produce for the configs[0]
backtest (oracle/gen_golden.py csv; data, not source).  vw needs no device; the estimators are -m gpu."""
import io
import json
import os

import numpy as np
import pandas as pd
import pytest

from incorporating_different_sources_amd import synthetic

from conftest import GOLDEN

FILES = ("simple_returns", "turnover", "portfolio_weights_metrics")
KEYS = {"simple_returns": "portfolio_simple_returns_series", "turnover": "portfolio_turnover_series",
        "portfolio_weights_metrics": "portfolio_weights_metrics_df"}


def _load():
    return json.load(open(os.path.join(GOLDEN, "backtest_k10_n60_daily_csv.json")))


def _read_like_main(text_or_path):
    """src/main.py:56-66 `pd.read_csv(file, index_col=0, parse_dates=True[, squeeze=True])` (squeeze left pandas in 2.0:
    `.squeeze("columns")` is its documented replacement)."""
    return pd.read_csv(text_or_path, index_col=0, parse_dates=True)


def _check(strat, tmp_path):
    from incorporating_different_sources_amd import portfolio_calculations as pc
    g = _load()
    md, _ = synthetic.make_market_data(n_tickers=g["n_tickers"], n_days=g["n_days"], seed=g["seed"])
    days = md["stock_prices_df"].index
    simple = strat in ("vw", "ew")
    spec = {"weighting_strategy": strat, "size": g["size"], "risk_aversion": None if simple else 5, "turnover_cost": 15,
            "rebalancing_frequency": g["rebal"], "rolling_window": g["N"], "rolling_window_frequency": g["window_freq"],
            "mcm_scaling": None if simple or strat == "jeffreys" else 1, "display_name": "Display " + strat}
    res = pc.backtest_portfolio(spec, days[g["start_idx"]], days[-1], md)
    for name in FILES:
        ref_text = g["files"][strat][name]
        obj = res[KEYS[name]]
        text = obj.to_csv(header=True)                                   # exactly the call of src/main.py:79-81
        ref_lines, lines = ref_text.splitlines(), text.splitlines()
        assert lines[0] == ref_lines[0], name                            # header: '' + Series name / the five columns
        assert len(lines) == len(ref_lines), name
        assert [l.split(",")[0] for l in lines] == [l.split(",")[0] for l in ref_lines], name     # index text, row by row
        ours, ref = _read_like_main(io.StringIO(text)), _read_like_main(io.StringIO(ref_text))
        assert list(ours.columns) == list(ref.columns) and ours.index.equals(ref.index)
        assert isinstance(ours.index, pd.DatetimeIndex)
        np.testing.assert_allclose(ours.to_numpy(), ref.to_numpy(), rtol=1e-10, atol=1e-15, equal_nan=True)
        assert np.array_equal(np.isnan(ours.to_numpy()), np.isnan(ref.to_numpy()))     # empty fields where the reference has them
        # the cache-load branch: what was written is what is read back next run
        path = tmp_path / f"{strat}_{name}.csv"
        obj.to_csv(path, header=True)
        back = _read_like_main(path)
        if name != "portfolio_weights_metrics":
            back = back.squeeze("columns")
            assert isinstance(back, pd.Series) and back.name == spec["display_name"] == obj.name
        assert back.index.equals(obj.index)
        # (read_csv's default "fast" float parser is not a round-trip parser: a few ulps)
        np.testing.assert_allclose(back.to_numpy(), obj.to_numpy(), rtol=1e-12, atol=1e-18, equal_nan=True)
        exact = pd.read_csv(path, index_col=0, parse_dates=True, float_precision="round_trip")
        np.testing.assert_array_equal(exact.to_numpy().ravel(), np.asarray(obj.to_numpy()).ravel())   # the TEXT is exact


def test_result_files_of_a_passive_backtest_match_the_references_text(tmp_path):
    _check("vw", tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("strat", ["conjugate_hf_vix_vw", "jeffreys"])
def test_result_files_match_the_references_text(strat, tmp_path):
    _check(strat, tmp_path)
