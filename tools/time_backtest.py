"""Where a whole `backtest_portfolio` call spends its time: batch packing + device solve vs the daily replay
loop (ref:1127-1219).  GPU box: python tools/time_backtest.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats
import numpy as np, pandas as pd
from incorporating_different_sources_amd import portfolio_calculations as pc, synthetic

md, tickers = synthetic.make_market_data(n_tickers=120, n_days=900, seed=1)
days = md["stock_prices_df"].index
spec = {"weighting_strategy": "conjugate_hf_vix_vw", "size": 100, "risk_aversion": 5, "turnover_cost": 15,
        "rebalancing_frequency": "daily", "rolling_window": 250, "rolling_window_frequency": "daily",
        "mcm_scaling": 1, "display_name": "conjugate"}
start, end = days[255], days[-1]
pc.backtest_portfolio(spec, days[-5], end, md)          # warm (library load, panel cache)
t0 = time.perf_counter()
dates = [pd.Timestamp(d) for d in days if start <= d <= end]
frames = pc.calculate_portfolio_weights_batch(pc.rebalancing_schedule(dates, "daily"), spec, md)
t1 = time.perf_counter()
res = pc.backtest_portfolio(spec, start, end, md)
t2 = time.perf_counter()
n = len(dates)
print(f"{n} trading days, k=100 of 120: weights for all dates (pack + upload + solve + download) {1e3 * (t1 - t0):.0f} ms "
      f"({1e3 * (t1 - t0) / n:.2f} ms/date); whole backtest {1e3 * (t2 - t1):.0f} ms ({1e3 * (t2 - t1) / n:.2f} ms/date)")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable(); pc.backtest_portfolio(spec, start, end, md); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
