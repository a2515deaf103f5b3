#!/bin/bash
# The round's evidence for the bench line, in one GPU call: bench JSON lines, rocprofv3 kernel stats of the
# same command, HBM traffic counters in separate --pmc passes (never combined with tracing).
# Usage on the GPU box: tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --strategy jeffreys --no-cpu-baseline --no-end-to-end > "$OUT/bench_c2j.json" 2>> "$OUT/bench.err"
python3 bench.py --config 1 --no-cpu-baseline --no-end-to-end > "$OUT/bench_c1.json" 2>> "$OUT/bench.err"
python3 bench.py --config 3 --windows 4096 --steps 5 --warmup 1 --no-end-to-end > "$OUT/bench_c3.json" 2>> "$OUT/bench.err"
python3 bench.py --config 5 --windows 384 --steps 3 --warmup 1 --no-end-to-end > "$OUT/bench_c5.json" 2>> "$OUT/bench.err"
python3 bench.py --rehearse-gather --no-cpu-baseline --no-end-to-end > "$OUT/bench_rehearse_gather.json" 2>> "$OUT/bench.err"
# configs[2] at its stated 50,000 windows (both priors; Jeffreys is singular at k=500 > N-2: throughput only) and configs[3]'s
# per-GPU shard (25,000 windows) with the overlapped RCCL gather on a one-rank communicator
python3 bench.py --config 3 --steps 3 --warmup 1 --no-end-to-end > "$OUT/bench_c3_50k.json" 2>> "$OUT/bench.err"
python3 bench.py --config 3 --strategy jeffreys --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline > "$OUT/bench_c3j_50k.json" 2>> "$OUT/bench.err"
python3 bench.py --config 4 --rehearse-gather --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > "$OUT/bench_c4_shard.json" 2>> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_c3" -- python3 "$ROOT/bench.py" --config 3 --windows 4096 --steps 5 --warmup 1 --no-end-to-end --no-cpu-baseline > /dev/null 2>> "$OUT/rocprof.err"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_c5" -- python3 "$ROOT/bench.py" --config 5 --windows 384 --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline > /dev/null 2>> "$OUT/rocprof.err"
rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>> "$OUT/rocprof.err"
rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>> "$OUT/rocprof.err"
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
ls -R "$OUT" | head -60
