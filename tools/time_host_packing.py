"""Diagnostic: host packing time per rebalancing date, batch-native (batch.pack_windows) vs the
frame-based packer that mirrors the reference's per-date slicing.  CPU only."""
import sys, os, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LOGGING_LEVEL", "WARNING")
from incorporating_different_sources_amd import batch, synthetic, portfolio_calculations as pc

n_t, n_d, size, N = int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 900, 50, 250
md, _ = synthetic.make_market_data(n_tickers=n_t, n_days=n_d, seed=1)
days = md["stock_prices_df"].index
for wf, rebal in (("daily", "daily"), ("weekly", "monthly")):
    Nw = N if wf == "daily" else 60
    spec = {"weighting_strategy": "conjugate_hf_vix_vw", "size": size, "risk_aversion": 5, "turnover_cost": 15,
            "rebalancing_frequency": rebal, "rolling_window": Nw, "rolling_window_frequency": wf, "mcm_scaling": 1,
            "display_name": "x"}
    first = Nw * (1 if wf == "daily" else 5) + 5
    dates = pc.rebalancing_schedule([pd.Timestamp(d) for d in days[first:]], rebal)
    t0 = time.perf_counter(); kw, labels = batch.pack_windows(dates, spec, md); t1 = time.perf_counter()
    sub = dates[:: max(1, len(dates) // 40)]
    t2 = time.perf_counter(); [pc._pack_window(d, spec, md) for d in sub]; t3 = time.perf_counter()
    print(f"{wf:7s} window / {rebal:7s} rebalancing, {n_t} tickers x {n_d} days, {len(dates)} dates: "
          f"batch {1e3 * (t1 - t0) / len(dates):.2f} ms/date (incl. one-off panel build), frame-based {1e3 * (t3 - t2) / len(sub):.2f} ms/date")
