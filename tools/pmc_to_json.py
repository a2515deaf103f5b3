#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/profile_round.sh into profiles/pmc_traffic.json (what bench.py attaches as
`roofline.traffic` and `roofline.executed`).  Usage: tools/pmc_to_json.py gpurun_out/prof_<tag> > profiles/pmc_traffic.json

Per launch = per bench step: the counters of every kernel of a step (block Grams, window sums, window kernel) summed.
FETCH_SIZE x 2 (gfx950 reports half the bytes of streaming reads; calibrated for this library's 8-byte-per-lane loads in
round 2, profiles/r02_fetch_size_calibration.csv), WRITE_SIZE x 1, both in KB."""
import collections
import csv
import glob
import json
import os
import sys


def counters(d):
    """{kernel short name: {counter: mean per dispatch}}, {kernel: dispatches} of one pass directory."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = next((s for s in ("posterior_wave2_kernel", "posterior_wave_kernel", "posterior_fused_kernel",
                                      "block_gram_kernel", "tp_window_sums_kernel") if s in name), None)
            if short:
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def kernel_ms(stats_dir):
    out = {}
    for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            short = next((s for s in ("posterior_wave2_kernel", "posterior_wave_kernel", "posterior_fused_kernel",
                                      "block_gram_kernel", "tp_window_sums_kernel") if s in r["Name"]), None)
            if short:
                out[short] = float(r["AverageNs"]) / 1e6
    return out


def bench_line(path):
    try:
        for line in open(path):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


def main():
    out_dir = sys.argv[1]
    fetch, write = counters(os.path.join(out_dir, "pmc_fetch")), counters(os.path.join(out_dir, "pmc_write"))
    base = bench_line(os.path.join(out_dir, "bench_under_rocprof.json")) or {}
    cfg = base.get("config", {})
    res = {
        "round": 3,
        "workload": cfg.get("workload", "?").replace(" assets", "").replace("-day window", "").replace(" per GPU", ""),
        "FETCH_SIZE_KB_per_launch_raw": {k: v.get("FETCH_SIZE") for k, v in fetch.items()},
        "WRITE_SIZE_KB_per_launch": {k: v.get("WRITE_SIZE") for k, v in write.items()},
        "correction": "FETCH_SIZE x 2, WRITE_SIZE x 1 (MI355X_MICROARCH.md HBM section; calibrated for 8 B/lane loads in "
                      "profiles/r02_fetch_size_calibration.csv)",
    }
    by_kernel = {}
    for k in set(fetch) | set(write):
        by_kernel[k] = 2048.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) + 1024.0 * write.get(k, {}).get("WRITE_SIZE", 0.0)
    res["hbm_bytes_per_launch_by_kernel"] = by_kernel
    res["hbm_bytes_per_launch"] = sum(by_kernel.values())
    W = cfg.get("windows_per_gpu")
    if W and base.get("roofline"):
        res["algorithmic_bytes_per_launch"] = base["roofline"]["alg_bytes_per_window"] * W
    sq = {}
    for layout in ("contiguous", "no-shared-gram", "index"):
        c = counters(os.path.join(out_dir, "pmc_sq_" + layout))
        line = bench_line(os.path.join(out_dir, f"bench_pmc_sq_{layout}.json"))
        if not c or not line:
            continue
        f2 = counters(os.path.join(out_dir, "pmc_fetch_" + layout))
        tot = collections.defaultdict(float)
        for cs in c.values():
            for name, v in cs.items():
                tot[name] += v
        stats = kernel_ms(os.path.join(out_dir, "stats" if layout == "contiguous" else "stats_" + layout))
        sq[layout] = dict(k=line["config"]["k"], windows=line["config"]["windows_per_gpu"], strategy=line["config"]["strategy"],
                          kernel_ms_under_pmc=line["roofline"]["kernel_ms"],
                          # the un-profiled duration of the same launches (rocprofv3 --kernel-trace --stats of the same command,
                          # averages summed over the kernels of a step): the base of the pipe-busy fraction
                          kernel_ms_rocprof=sum(stats.values()) if stats else None, kernels=sorted(c),
                          hbm_read_bytes_per_launch=sum(2048.0 * v.get("FETCH_SIZE", 0.0) for v in f2.values()) or None,
                          source=f"rocprofv3 --pmc passes of `bench.py --layout {layout} --steps 3 --warmup 1` "
                                 f"(tools/profile_round.sh), counters summed over the kernels of one step",
                          **{name: tot[name] for name in sorted(tot)})
    res["sq_counters"] = sq
    res["kernel_ms_rocprof"] = {"contiguous": kernel_ms(os.path.join(out_dir, "stats")),
                                "no-shared-gram": kernel_ms(os.path.join(out_dir, "stats_no-shared-gram")),
                                "index": kernel_ms(os.path.join(out_dir, "stats_index"))}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
