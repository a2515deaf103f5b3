#!/usr/bin/env python3
"""bench.py - rolling windows/sec of the posterior hot path on N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its own N worker processes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the posterior kernels over one batch of synthetic windows resident in HBM.
N = 1: BASELINE.json configs[1] (k=100 assets, n=250-day window, 10k windows, conjugate prior with a 78-bar intraday
scatter and a VIX-style n0).  N > 1: configs[3], the same shapes as 200k windows over 8 GPUs = 25k windows per rank
(weak scaling: every rank owns its own 25k windows at every N); a step ends with the RCCL gather of the weights to
rank 0 (`--config 5`: k=1000, n=500, 125k windows per rank).  Rank 0 prints ONE JSON line.

An RCCL failure is fatal (exit code != 0): there is no other transport in the timed path.  `--rehearse-world N`
rehearses the N-rank HOST logic on a box with fewer GPUs (ranks share devices, RCCL is skipped by construction and
the JSON line says so); without it, fewer GPUs than ranks is refused.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CLOCK_HZ = 2.4e9               # nominal shader clock the peak is quoted at
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
FLOPS_PER_MFMA_F64 = 2048      # v_mfma_f64_16x16x4_f64: 16 x 16 x 4 multiply-adds
WINDOWS_PER_RANK = {4: 25_000, 5: 125_000}     # configs[3] / configs[4]: 200k resp. 1M windows over 8 GPUs
HF_PANEL_BUDGET_BYTES = 24e9   # host bytes of the synthetic intraday panels of ALL ranks above which they wrap (hf_period)
# the unmodified reference, one window per call, measured in the survey container (8 cores)
REFERENCE_AS_SHIPPED = {100: {"windows_per_s": 29.0, "check_off_windows_per_s": 74.0},
                        10: {"windows_per_s": 52.0, "check_off_windows_per_s": 84.0},
                        500: {"windows_per_s": 5.8, "check_off_windows_per_s": 33.0},
                        1000: {"windows_per_s": 1.0, "check_off_windows_per_s": 15.0}}


def alg_bytes_per_window(k, n_r, m, conj=True):
    """read X, Y, w0; write k weights; 4 scalars."""
    return 8 * ((n_r + m) * k + 2 * k + 4) if conj else 8 * (n_r * k + k + 3)


def alg_flops_per_window(k, n_r, m, conj=True):
    """symmetric-half Grams, Cholesky, S0 w0, two triangular solves, one quadratic form."""
    return ((n_r + m) * k * (k + 1) + k ** 3 / 3 + 6 * k ** 2) if conj else (n_r * k * (k + 1) + k ** 3 / 3 + 5 * k ** 2)


def register_tile_kernel(launch):
    """Which register-tile kernel the library ran, from the launch geometry it reports (`tp_last_launch`): one wavefront
    per window (64 threads), two (128, csrc/posterior_wave2_impl.h) or the multi-wave kernel."""
    return {64: "posterior_wave_kernel (one wavefront per window)",
            128: "posterior_wave2_kernel (two wavefronts per window)"}.get(launch["block"], "posterior_fused_kernel")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0,
                    help="BASELINE.json config id (default: 2 = k100/n250/10k windows at N=1, 4 = its 8-GPU form at N>1)")
    ap.add_argument("--windows", type=int, default=0, help="override windows per GPU")
    ap.add_argument("--strategy", default="conjugate", choices=["conjugate", "jeffreys"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive legs (N=1 only, after the timed region)")
    ap.add_argument("--rehearse-gather", action="store_true",
                    help="N=1 only: run the N>1 step (RCCL gather on the second stream + its verification) on a "
                         "one-rank communicator")
    ap.add_argument("--layout", default="contiguous", choices=["contiguous", "no-shared-gram", "index"],
                    help="data layout of the TIMED batch: contiguous rolling windows (default: the headline; the daily "
                         "Gram of whole row blocks comes from sums all windows share), the same without sharing "
                         "(TP_FLAG_NO_SHARED_GRAM), or the index layout (row_idx / col_idx / rf_adj / hf_row_idx: what "
                         "batch.pack_windows produces for real backtests with changing universes, ref:611-658, 149-156)")
    ap.add_argument("--no-general-layout", action="store_true",
                    help="skip the no-shared-gram and index-layout legs that follow the timed region (N=1 only)")
    ap.add_argument("--hf-period", type=int, default=-1,
                    help="days after which the synthetic intraday panel wraps (0: never; default: never unless the panel "
                         "would exceed 24 GB of host memory per rank)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="tp_set_option on the handle before anything runs (A/B measurements), e.g. --opt tiled_lanes=3 "
                         "--opt tiled_arena_mib=64 --opt wave_kernel=0")
    ap.add_argument("--k", type=int, default=0, help="override the universe size of the config (N, n_r, m stay): sweeps")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="run the host logic of this many ranks on however many GPUs the box has (ranks share devices, "
                         "no RCCL, host gather): a rehearsal, never a measurement")
    return ap.parse_args()


def cpu_quota_cores():
    """CPU cores this job's cgroup may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            return float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def visible_gpus_sysfs():
    """GPUs the KFD topology lists, read from sysfs: the launcher must not initialise HIP (its workers do)."""
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except OSError:
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var):
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip() != ""]))
    return n


def spawn_workers(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N worker processes of this script, one per GPU, BEFORE
    anything in this process touches the GPU, and wait for them.  Rank 0's stdout is this process's stdout."""
    world = args.rehearse_world or args.gpus
    have = visible_gpus_sysfs()
    if not args.rehearse_world and have is not None and have < world:
        print(f"bench.py: --gpus {world} needs {world} GPUs, this box shows {have} (one rank per GPU; "
              f"--rehearse-world {world} rehearses the host logic without RCCL)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TP_CONTROL_PORT=str(port), TP_BENCH_WORKER="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(world))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()          # exactly the processes started above
        time.sleep(0.05)
    return rc


def worker(args):
    # Libraries underneath (RCCL) print banners on the C-level stdout.  The contract is ONE JSON
    # line on stdout, so everything else is routed to stderr and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    from incorporating_different_sources_amd import _native, shard, synthetic

    cp = shard.ControlPlane()
    want_world = args.rehearse_world or args.gpus
    if cp.world != want_world:
        if cp.rank == 0:
            print(f"bench.py: --gpus {want_world} but WORLD_SIZE={cp.world}", file=sys.stderr)
        sys.exit(2)

    ndev = _native.device_count()
    rehearsal = bool(args.rehearse_world)
    if ndev < 1:
        print("bench.py: no HIP device (there is no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    if not rehearsal and cp.world > ndev:
        # one rank per GPU, never two ranks on one device: RCCL rejects that by design
        print(f"[rank {cp.rank}] bench.py: {cp.world} ranks but {ndev} visible GPU(s); refusing "
              f"(--rehearse-world {cp.world} rehearses the host logic without RCCL)", file=sys.stderr)
        sys.exit(3)

    config = args.config or (2 if cp.world == 1 else 4)
    shp = synthetic.config_shapes(config)
    if args.k:
        shp = dict(shp, k=args.k)
    k, N, n_r, m = shp["k"], shp["N"], shp["n_r"], shp["m"]
    W = args.windows or (WINDOWS_PER_RANK.get(config, shp["W"]) if cp.world > 1 else min(shp["W"], WINDOWS_PER_RANK.get(config, shp["W"])))
    conj = args.strategy == "conjugate"
    # rank r owns its own windows (weak scaling): an independent synthetic panel per rank
    # intraday panel: one row block per window day, un-wrapped whenever it fits the host (a wrapped panel is a
    # friendlier HBM footprint than SURVEY section 8(d)'s shapes)
    hf_period = args.hf_period
    if hf_period < 0:
        full_bytes = (W + shp["hf_days"] - 1) * synthetic.BARS_PER_DAY * k * 8.0
        budget = HF_PANEL_BUDGET_BYTES / cp.world            # the ranks of a node share its host memory
        hf_period = 0 if (full_bytes <= budget or not conj) else \
            max(1, int(budget / (synthetic.BARS_PER_DAY * k * 8.0)) - shp["hf_days"])
    inp = synthetic.make_kernel_inputs(k, N, W, seed=shp["seed"] + 1000 * cp.rank, hf_days=shp["hf_days"],
                                       hf_period=hf_period)

    dev = _native.Device(cp.local_rank % ndev if rehearsal else cp.local_rank)
    for item in args.opt:
        name, _, val = item.partition("=")
        dev.set_option(name.strip(), int(val))
    gather_mode, rccl_ranks = "none", None
    if (cp.world == 1 and args.rehearse_gather) or (cp.world > 1 and not rehearsal):
        shard.init_rccl(dev, cp)            # raises -> non-zero exit: an RCCL failure is never papered over
        gather_mode = "rccl"
        rccl_ranks = dev.comm_count()
        if rccl_ranks != cp.world:
            print(f"[rank {cp.rank}] communicator has {rccl_ranks} ranks, expected {cp.world}", file=sys.stderr)
            sys.exit(4)
    elif cp.world > 1:
        gather_mode = "host-tcp (rehearsal: ranks share devices, no RCCL)"

    kw = dict(panel=inp["panel"], start=inp["start"])
    if conj:
        kw.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"], w0=inp["w0"], n0=inp["n0"])

    def make_batch(layout):
        """The same W windows in one of the three layouts (all resident in HBM before anything is timed)."""
        flags = _native.FLAG_NO_SHARED_GRAM if layout == "no-shared-gram" else 0
        b = dev.batch(args.strategy, k, N, n_r, 5.0, W, m if conj else 0, flags=flags)
        if layout != "index":
            b.upload(**kw)
            return b
        # identity index arrays: exactly the windows of the contiguous layout, addressed the way batch.pack_windows
        # addresses real ones (explicit rows, gathered columns, a per-row risk-free adjustment)
        ikw = dict(panel=inp["panel"], row_idx=(inp["start"][:, None] + np.arange(n_r)[None, :]).astype(np.int32),
                   col_idx=np.tile(np.arange(k, dtype=np.int32), (W, 1)), rf_adj=np.zeros((W, n_r)))
        if conj:
            ikw.update(hf_panel=inp["hf_panel"], hf_row_idx=(inp["hf_start"][:, None] + np.arange(m)[None, :]).astype(np.int32),
                       w0=inp["w0"], n0=inp["n0"])
        b.upload(**ikw)
        return b

    batch = make_batch(args.layout)
    h2d_ms = dev.last_timing()["h2d_ms"]
    shared_blocks = batch.shared_gram_blocks()
    shared_hf = batch.shared_intraday_blocks()

    def step():
        batch.run()
        if gather_mode == "rccl":
            batch.gather_async(root=0)   # on the gather stream: overlaps the next step's kernel; the gathered
                                         # weights stay in rank 0's HBM, like N=1; dev.synchronize() waits for it
        elif cp.world > 1:               # rehearsal only
            wts, st, _ = batch.download(want_aux=False)
            cp.gather_host(wts, root=0)

    for _ in range(args.warmup):
        step()
    dev.synchronize()
    cp.barrier()
    t0 = time.perf_counter()
    dev.region_begin()                   # HIP events on the kernel stream: the region, and one pair per step's launches
    for _ in range(args.steps):
        step()
    region_ms = dev.region_end()
    dev.synchronize()
    cp.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = cp.max(elapsed)
    tim = dev.last_timing()
    steps_ms = np.sort(dev.region_steps())
    # N = 1: the region holds nothing but the steps' kernels back to back; N > 1: the kernels only (the gather runs on
    # its own stream), so the mean of the per-step brackets
    kernel_ms = (region_ms / args.steps) if cp.world == 1 else float(steps_ms.mean()) if len(steps_ms) else tim["kernel_ms"]
    step_stats = None
    if len(steps_ms):
        step_stats = {"n": int(len(steps_ms)), "median": float(np.median(steps_ms)), "min": float(steps_ms[0]),
                      "max": float(steps_ms[-1]), "p90": float(steps_ms[int(0.9 * (len(steps_ms) - 1))]),
                      "source": "one HIP-event pair per step on the kernel stream (tp_region_steps)"}

    weights, status, aux = batch.download()
    d2h_ms = dev.last_timing()["d2h_ms"]
    gathered_ok = None
    if gather_mode == "rccl":
        # after the timed region: check the collective - every rank's slice on root against a checksum of
        # what that rank computed (control plane), rank 0's slice element by element
        sums = cp.gather_host(np.array([weights.sum(), np.abs(weights).sum(), float(status.sum())]), root=0)
        if cp.rank == 0:
            wall, sall = batch.download_gathered()
            gathered_ok = bool(np.array_equal(wall[0], weights) and np.array_equal(sall[0], status)
                               and np.isfinite(wall).all()
                               and all(wall[r].sum() == sums[r][0] and np.abs(wall[r]).sum() == sums[r][1]
                                       and float(sall[r].sum()) == sums[r][2] for r in range(cp.world)))
            if not gathered_ok:
                print("bench.py: the gathered weights do not match what the ranks computed", file=sys.stderr)
        ok_all = cp.max(0.0 if (cp.rank != 0 or gathered_ok) else 1.0)
        if ok_all > 0.0:
            sys.exit(5)
    n_bad = int(cp.sum(float((status != 0).sum())))
    launch = dev.last_launch()
    info = dev.info()

    # parity spot check against the CPU oracle on the first windows of this rank (not timed)
    from oracle import oracle

    def oracle_kw(n):
        o = dict(panel=inp["panel"], start=inp["start"][:n], n_r=n_r)
        if conj:
            o.update(hf_panel=inp["hf_panel"], hf_start=inp["hf_start"][:n], m=m, w0=inp["w0"][:n], n0=inp["n0"][:n])
        return o

    ns = min(W, 64 if k <= 239 else 8)
    ref, rstat, _ = oracle.posterior_batch_c(args.strategy, k, N, 5.0, **oracle_kw(ns))
    okw = (status[:ns] == 0) & (rstat == 0)
    parity = cp.max(float(np.abs(weights[:ns][okw] - ref[okw]).max()) if okw.any() else 0.0)

    cpu = cpu1 = cpu16 = None
    if cp.rank == 0 and not args.no_cpu_baseline:
        # the oracle's C restatement (OpenMP over windows) on a bounded sample of the same workload
        def time_cpu(threads, sample, budget_s):
            oracle.posterior_batch_c(args.strategy, k, N, 5.0, threads=threads, **oracle_kw(min(sample, 2 * threads)))
            reps, cdt = 0, 0.0
            c0 = time.perf_counter()
            while cdt < budget_s and reps < 400:
                oracle.posterior_batch_c(args.strategy, k, N, 5.0, threads=threads, **oracle_kw(sample))
                reps += 1
                cdt = time.perf_counter() - c0
            return {"value": sample * reps / cdt, "unit": "windows/s", "cores": threads, "kind": "port",
                    "sample": f"{reps} x {sample} windows of the same workload (oracle/tangency_oracle.c, OpenMP over "
                              f"windows), {cdt:.2f} s wall"}
        # all host cores and one core; the 16-thread leg is this pool's per-GPU CPU share
        per_thread = max(8, int(2.5e9 / alg_flops_per_window(k, n_r, m, conj)))      # ~1 s of one core
        all_cores = max(1, min(oracle.c_num_threads(), len(os.sched_getaffinity(0))))
        cpu = time_cpu(all_cores, min(W, per_thread * all_cores), 6.0)        # ~16 s of CPU work in the three legs
        cpu16 = time_cpu(min(16, all_cores), min(W, per_thread * min(16, all_cores)), 6.0) if all_cores > 16 else cpu
        cpu1 = time_cpu(1, min(W, per_thread), 4.0)
        shipped = REFERENCE_AS_SHIPPED.get(k)
        cpu["reference_as_shipped"] = None if shipped is None or not conj else dict(
            shipped, cores=8, unit="windows/s", source="BASELINE.md section 2: the unmodified reference "
            "(calculate_conjugate_hf_mcm_portfolio, one window per call) in the survey container; not re-measured here "
            "(the reference's Python does not travel to the GPU box)")

    # General layouts (N = 1, after the timed region, same windows): the headline's contiguous layout lets every window
    # take its whole 16-row blocks from Gram sums all windows share; a real backtest (changing universes ref:611-658,
    # resampled windows ref:149-156) arrives in the index layout, which pushes every row through the MFMAs.
    general = None
    if cp.world == 1 and not args.no_general_layout and args.layout == "contiguous":
        general = {}
        for layout in ("no-shared-gram", "index"):
            gb = make_batch(layout)
            gsteps = max(3, min(args.steps, 20))
            for _ in range(2):
                gb.run()
            dev.synchronize()
            g0 = time.perf_counter()
            dev.region_begin()
            for _ in range(gsteps):
                gb.run()
            g_ms = dev.region_end() / gsteps
            dev.synchronize()
            g_wall = (time.perf_counter() - g0) / gsteps
            gs = np.sort(dev.region_steps())
            gw, gst, _ = gb.download(want_aux=False)
            g_launch = dev.last_launch()
            g_tf = alg_flops_per_window(k, n_r, m, conj) * W / (g_ms * 1e-3) / 1e12
            general[layout.replace("-", "_")] = {
                "windows_per_s": W / (g_ms * 1e-3), "wall_windows_per_s": W / g_wall, "kernel_ms": g_ms, "step_ms_median": float(np.median(gs)) if len(gs) else None,
                "steps": gsteps, "shared_gram_row_blocks": gb.shared_gram_blocks(),
                "roofline_frac": g_tf / FP64_MFMA_PEAK_TFLOPS, "achieved_tflops": g_tf,
                "kernel": register_tile_kernel(g_launch) if k <= 239 else "tiled pipeline",
                "max_abs_diff_vs_headline": float(np.abs(gw - weights).max()),
                "windows_with_nonzero_status": int((gst != 0).sum()),
                "executed": executed_from_pmc(layout, k, W, args.strategy, g_ms)}
            gb.close()

    # PCIe-inclusive legs (never `value`): host buffers in, host buffers out
    e2e = None
    if cp.world == 1 and not args.no_end_to_end and k <= 239:
        e2e = end_to_end(dev, _native, args.strategy, k, N, n_r, m, W, conj, kw, weights)

    # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot run inside this process);
    # attached only when the profiled workload is the one benched
    traffic = None
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        if (f"k={k}," in pmc["workload"] and f"{W} windows" in pmc["workload"] and args.strategy in pmc["workload"]
                and args.layout == "contiguous"):
            traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        traffic = None

    if cp.rank == 0:
        total_windows = W * cp.world * args.steps
        value = total_windows / elapsed
        flops = alg_flops_per_window(k, n_r, m, conj) * W
        byts = alg_bytes_per_window(k, n_r, m, conj) * W
        ach_tf = flops / (kernel_ms * 1e-3) / 1e12
        ach_gbs = byts / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "rolling windows/sec",
            "value": value,
            "unit": "windows/s",
            "n_gpus": cp.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{config - 1}]" if not args.k else f"shapes of configs[{config - 1}] at another universe size")
                                   + f": k={k} assets, n={N}-day window, "
                                   f"{W} windows per GPU, {args.strategy} prior"
                                   + (f" (m={m} intraday returns, VIX-style n0)" if conj else ""),
                       "k": k, "N": N, "n_r": n_r, "m": m if conj else 0, "windows_per_gpu": W,
                       "strategy": args.strategy, "parallelism": f"windows sharded x{cp.world}",
                       "layout": args.layout, "hf_period_days": hf_period, "options": args.opt,
                       "shared_gram_row_blocks": shared_blocks, "shared_intraday_blocks_per_window": shared_hf,
                       "gather": gather_mode, "rccl_ranks": rccl_ranks, "gather_verified": gathered_ok,
                       "rehearsal": rehearsal, "seed": shp["seed"]},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_MFMA_PEAK_TFLOPS,
                         # what `frac` is: SURVEY section 8(d)'s ALGORITHMIC flops (every window its own full Grams) over the
                         # measured time.  With shared Gram sums the kernels EXECUTE fewer flops than that, so this is an
                         # algorithmic-equivalent rate, not matrix-pipe utilisation - `executed` carries that
                         "frac_kind": ("algorithmic-equivalent (shared Gram sums: executed flops < algorithmic flops"
                                       + (f"; the intraday Grams too, {shared_hf} of a window's days from shared block Grams: this rate "
                                          "can exceed the MFMA peak and is NOT a utilisation)" if shared_hf else ")"))
                                      if (shared_blocks or shared_hf) else "algorithmic flops; every row of every window goes through the MFMAs",
                         "executed": executed_from_pmc(args.layout, k, W, args.strategy, kernel_ms),
                         "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes)" if traffic else None,
                         "kernel": (register_tile_kernel(launch) + (" (+ block_gram_kernel and tp_window_sums_kernel in front of it, every step)" if shared_blocks else ""))
                                   if k <= 239 else ("tiled pipeline (intraday block Grams + sums + prefix + gram + diag / TRSM / SYRK + solve)" if shared_hf else "tiled pipeline (prior + prefix + gram + diag / TRSM / SYRK + solve)"),
                         "kernel_ms": kernel_ms,
                         "step_ms": step_stats,
                         "alg_flops_per_window": alg_flops_per_window(k, n_r, m, conj),
                         "alg_bytes_per_window": alg_bytes_per_window(k, n_r, m, conj)},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach_gbs / HBM_PEAK_GBS},
            "general_layout": general,
            "cpu_baseline": cpu,
            "cpu_baseline_16t": cpu16,
            "cpu_baseline_1t": cpu1,
            "end_to_end": e2e,
            "end_to_end_windows_per_s": e2e["pinned_windows_per_s"] if e2e else None,
            "parity_max_abs_diff_vs_oracle": parity,
            "windows_with_nonzero_status": n_bad,
            "launch": launch,
            "device": info["name"].strip() or "AMD Instinct MI355X",
            "h2d_ms": h2d_ms, "d2h_ms": d2h_ms, "gather_ms": tim["gather_ms"] if gather_mode == "rccl" else None,
            "host": {"cpus": len(os.sched_getaffinity(0)), "cpu_quota_cores": cpu_quota_cores(),
                     "note": "a one-GPU box of this pool shares its host: the affinity mask shows every hardware thread, the "
                             "cgroup quota (when set) is what this job may actually use - the all-core CPU leg oversubscribes it"},
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    batch.close()
    dev.close()
    cp.close()


def executed_from_pmc(layout, k, W, strategy, kernel_ms):
    """Executed-work fractions of one step from the committed rocprofv3 --pmc passes of the same workload and layout
    (profiles/pmc_traffic.json "sq_counters"; counters cannot be read from inside this process): the flops the MFMA
    instructions actually performed over the measured time, and the share of SIMD-cycles the matrix pipe was busy during
    the profiled launches.  None when no pass of this workload / layout is committed."""
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        ent = pmc["sq_counters"][layout]
        if ent["k"] != k or ent["windows"] != W or ent["strategy"] != strategy:
            return None
        insts, busy = ent["SQ_INSTS_MFMA"], ent["SQ_VALU_MFMA_BUSY_CYCLES"]
        pmc_ms = ent.get("kernel_ms_rocprof") or kernel_ms      # duration of the same launches without counters attached
        ex_tf = insts * FLOPS_PER_MFMA_F64 / (kernel_ms * 1e-3) / 1e12
        return {"mfma_instructions_per_step": insts, "executed_tflops": ex_tf,
                "executed_mfma_frac": ex_tf / FP64_MFMA_PEAK_TFLOPS,
                "mfma_pipe_busy_frac": busy / (N_SIMD * pmc_ms * 1e-3 * CLOCK_HZ),
                "source": ent.get("source", "profiles/pmc_traffic.json"),
                "note": "SQ_INSTS_MFMA x 2048 flop / this run's kernel time / peak; SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                        "the rocprofv3 --stats kernel time of the same command x 2.4 GHz)"}
    except Exception:
        return None


def end_to_end(dev, _native, strategy, k, N, n_r, m, W, conj, kw, expect):
    """Host buffers in, host buffers out (H2D + kernel + D2H), three ways: pageable numpy arrays; page-locked arrays
    (`tp_host_alloc`); page-locked and STREAMED - two resident batches, the upload of batch i+1 on the copy stream
    under the kernel of batch i (`tp_batch_upload_async`).  Windows/s of each, never the headline `value`."""
    import numpy as np
    reps = 5
    out = {}

    def once(b, kwargs, outs):
        t0 = time.perf_counter()
        b.upload(**kwargs)
        b.run()
        b.download(want_aux=False, out=outs)
        return time.perf_counter() - t0

    b0 = dev.batch(strategy, k, N, n_r, 5.0, W, m if conj else 0)
    # pageable
    page_out = (np.empty((W, k)), np.empty(W, np.int32))
    once(b0, kw, page_out)
    ts = sorted(once(b0, kw, page_out) for _ in range(reps))
    out["pageable_windows_per_s"] = W / ts[len(ts) // 2]
    # page-locked
    pkw = {key: (_native.pinned_copy(v) if isinstance(v, np.ndarray) else v) for key, v in kw.items()}
    pin_out = (_native.pinned_empty((W, k)), _native.pinned_empty((W,), np.int32))
    once(b0, pkw, pin_out)
    ts = sorted(once(b0, pkw, pin_out) for _ in range(reps))
    t_seq = ts[len(ts) // 2]
    out["pinned_windows_per_s"] = W / t_seq
    out["pinned_ms"] = {"h2d": dev.last_timing()["h2d_ms"], "kernel": dev.last_timing()["kernel_ms"],
                        "d2h": dev.last_timing()["d2h_ms"], "total": t_seq * 1e3}
    ok = bool(np.array_equal(pin_out[0], expect))
    # streamed: upload of the next batch under the kernel of the current one
    b1 = dev.batch(strategy, k, N, n_r, 5.0, W, m if conj else 0)
    bs = [b0, b1]
    pin_out2 = (_native.pinned_empty((W, k)), _native.pinned_empty((W,), np.int32))
    outs = [pin_out, pin_out2]
    S = 8
    b0.upload_async(**pkw)
    t0 = time.perf_counter()
    for i in range(S):
        if i + 1 < S:
            bs[(i + 1) % 2].upload_async(**pkw)          # queued on the copy stream
        bs[i % 2].run()                                   # waits for ITS upload on the device
        bs[i % 2].download(want_aux=False, out=outs[i % 2])
    dev.synchronize()
    t_stream = (time.perf_counter() - t0) / S
    for b in bs:
        b.upload_wait()
    out["streamed_windows_per_s"] = W / t_stream
    out["streamed_ms_per_batch"] = t_stream * 1e3
    out["stream_overlap"] = max(0.0, 1.0 - t_stream / t_seq)    # share of the sequential time hidden by the overlap
    out["results_identical"] = ok and bool(np.array_equal(pin_out2[0], expect) and np.array_equal(pin_out[0], expect))
    out["note"] = "PCIe-inclusive; inputs are dominated by the intraday panel (one 78-bar day per window)"
    b0.close()
    b1.close()
    return out


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and (args.gpus > 1 or args.rehearse_world > 1):
        sys.exit(spawn_workers(args))
    worker(args)


if __name__ == "__main__":
    main()
