#!/bin/bash
# The round's evidence for the bench line, in one GPU call: bench JSON lines, rocprofv3 kernel stats of the
# same command, HBM traffic and matrix-pipe counters in separate --pmc passes (never combined with tracing).
# Usage on the GPU box: tools/profile_round.sh <tag> [quick]    -> gpurun_out/prof_<tag>/
# `quick`: the configs[1] line, its kernel stats and PMC passes only (no k=500 / k=1000 / shard lines).
set -e
TAG=${1:-r03}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"
if [ -z "$QUICK" ]; then
python3 bench.py --strategy jeffreys --no-cpu-baseline --no-end-to-end > "$OUT/bench_c2j.json" 2>> "$OUT/bench.err"
python3 bench.py --config 1 --no-cpu-baseline --no-end-to-end > "$OUT/bench_c1.json" 2>> "$OUT/bench.err"
python3 bench.py --config 3 --windows 4096 --steps 5 --warmup 1 --no-end-to-end > "$OUT/bench_c3.json" 2>> "$OUT/bench.err"
python3 bench.py --config 5 --windows 384 --steps 3 --warmup 1 --no-end-to-end > "$OUT/bench_c5.json" 2>> "$OUT/bench.err"
python3 bench.py --config 5 --windows 4096 --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-general-layout > "$OUT/bench_c5_4096_windows.json" 2>> "$OUT/bench.err"
echo "c3 / c5 done"
# configs[2] at its stated 50,000 windows with the intraday panel NOT wrapped (15.6 GB), Jeffreys there for throughput
# (singular at k=500 > N-2), and configs[3]'s per-GPU shard with the overlapped RCCL gather on a one-rank communicator
python3 bench.py --config 3 --steps 3 --warmup 1 --no-end-to-end --no-general-layout > "$OUT/bench_c3_50k.json" 2>> "$OUT/bench.err"
echo "c3 50k done"
python3 bench.py --config 3 --strategy jeffreys --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --no-general-layout > "$OUT/bench_c3j_50k.json" 2>> "$OUT/bench.err"
python3 bench.py --config 4 --rehearse-gather --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > "$OUT/bench_c4_shard.json" 2>> "$OUT/bench.err"
fi
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
Q="--no-cpu-baseline --no-end-to-end --no-general-layout"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -- python3 $B --steps 20 --warmup 3 $Q > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
for L in no-shared-gram index; do
  rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_$L" -- python3 $B --layout $L --steps 20 --warmup 3 $Q > "$OUT/bench_${L}_under_rocprof.json" 2>> "$OUT/rocprof.err"
done
echo "kernel stats done"
if [ -z "$QUICK" ]; then
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_c3" -- python3 $B --config 3 --windows 4096 --steps 5 --warmup 1 $Q > /dev/null 2>> "$OUT/rocprof.err"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_c5" -- python3 $B --config 5 --windows 384 --steps 3 --warmup 1 $Q > /dev/null 2>> "$OUT/rocprof.err"
fi
rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -- python3 $B --steps 3 --warmup 1 $Q > /dev/null 2>> "$OUT/rocprof.err"
rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -- python3 $B --steps 3 --warmup 1 $Q > /dev/null 2>> "$OUT/rocprof.err"
# matrix-pipe counters per layout: executed MFMA instructions and pipe-busy cycles (bench.py's roofline.executed)
for L in contiguous no-shared-gram index; do
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
    -f csv -d "$OUT/pmc_sq_$L" -- python3 $B --layout $L --steps 3 --warmup 1 $Q > "$OUT/bench_pmc_sq_$L.json" 2>> "$OUT/rocprof.err"
  rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch_$L" -- python3 $B --layout $L --steps 3 --warmup 1 $Q > /dev/null 2>> "$OUT/rocprof.err"
done
echo "pmc done"
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
python3 "$ROOT/tools/pmc_to_json.py" "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_to_json.err" || echo "pmc_to_json failed"
ls "$OUT"
