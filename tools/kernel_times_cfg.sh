#!/bin/bash
# per-kernel totals of one bench run under rocprofv3.  usage (GPU box): tools/kernel_times_cfg.sh <tag> <bench.py args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/kst_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/kst_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    print("%-60s calls %5s avg %10.1f us  share %5.1f%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
