#!/bin/bash
# Stall / utilisation counters of the fused kernel, one rocprofv3 --pmc pass per counter group (never
# combined with tracing).  Usage on the GPU box: tools/pmc_passes.sh <outdir under gpurun_out> [kernel substring] [bench.py args]
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-pmc_stall}
KERNEL=${2:-posterior_wave_kernel}
BENCH_ARGS=${3:---steps 2 --warmup 1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES" \
  "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -f csv -d "$OUT/pass$i" -- python3 "${GRAFT_REPO_ROOT:-$PWD}/bench.py" $BENCH_ARGS --no-cpu-baseline --no-end-to-end > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; }
done
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); fh.write(line + "\n")
PY
