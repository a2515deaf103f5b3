#!/bin/bash
# Dynamic instruction counts per window of the fused kernel for one or more builds of the library
# (rocprofv3 --pmc, no tracing).  Usage on the GPU box: tools/pmc_insts.sh libA.so [libB.so ...]
# With a stamp build, TP_PHASE_LIMIT=1 in the environment restricts the kernel to its Gram phases.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_insts
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export TANGENCY_LIB=$ROOT/incorporating_different_sources_amd/$lib
  rm -rf "$OUT/$lib"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -f csv -d "$OUT/$lib" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/$lib.log" 2>&1 || { echo "pass failed"; tail -5 "$OUT/$lib.log"; }
  python3 - "$OUT/$lib" "$lib" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "posterior_fused_kernel" in r["Kernel_Name"] or "posterior_wave_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 10000, 1) for k, v in sorted(acc.items())})
PY
done
