"""Host-side cost of the calls of one overlapped step (run + gather_async) on a one-rank communicator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from incorporating_different_sources_amd import _native, synthetic
shp = synthetic.config_shapes(2)
inp = synthetic.make_kernel_inputs(shp["k"], shp["N"], shp["W"], seed=1)
dev = _native.Device(0)
dev.comm_init(_native.Device.comm_unique_id(), 0, 1)
b = dev.batch("conjugate", shp["k"], shp["N"], shp["n_r"], 5.0, shp["W"], shp["m"])
b.upload(panel=inp["panel"], start=inp["start"], hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])
for _ in range(3):
    b.run(); b.gather_async(0)
dev.synchronize()
tr, tg = [], []
t_all = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter(); b.run(); t1 = time.perf_counter(); b.gather_async(0); t2 = time.perf_counter()
    tr.append(t1 - t0); tg.append(t2 - t1)
t_issue = time.perf_counter() - t_all
dev.synchronize()
t_done = time.perf_counter() - t_all
print(f"host: run {1e6 * np.median(tr):.0f} us, gather_async {1e6 * np.median(tg):.0f} us (max {1e6 * max(tg):.0f}); "
      f"20 steps issued in {1e3 * t_issue:.2f} ms, finished in {1e3 * t_done:.2f} ms")
print("gather_async per step (us):", [int(1e6 * x) for x in tg])
