#!/bin/bash
# Diagnostic: bench several builds of libtangency in one GPU call.  usage: tools/compare_libs.sh lib1.so lib2.so ...
for L in "$@"; do
  echo "== $L"
  TANGENCY_LIB=$PWD/incorporating_different_sources_amd/$L timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['parity_max_abs_diff_vs_oracle'], d['launch'])"
done
