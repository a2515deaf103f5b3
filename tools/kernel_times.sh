#!/bin/bash
# per-kernel average times of a bench run under rocprofv3 for a given library
L=$1
cd /tmp && export TMPDIR=/tmp
TANGENCY_LIB=$GRAFT_REPO_ROOT/incorporating_different_sources_amd/$L rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/kst_$L -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/kst_$L/**/*kernel_stats.csv", recursive=True)[0]
print("$L", [(r["Name"].split("::")[-1][:22], round(float(r["AverageNs"])/1e3,1)) for r in csv.DictReader(open(f))])
PY
