#!/bin/bash
# On-chip range of the `_tc` kernels: batches (pipeline 0) / rolling prefetch (2) x lookahead x LDS levels unrolled,
# default depth ladder, K = 160 / 80 / 60, three processes per setting, ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for cfg in "0 0 0" "2 4 0" "2 6 0" "2 4 1" "2 6 1" "2 3 1"; do
  set -- $cfg
  echo -n "pipeline=$1 lookahead=$2 unroll_lds=$3  vadv K=160/80/60: "
  for rep in 1 2 3; do
    GT4MI_CODEGEN_TOP_CACHE_PIPELINE=$1 GT4MI_CODEGEN_TOP_CACHE_LOOKAHEAD=$2 GT4MI_CODEGEN_TOP_CACHE_UNROLL_LDS=$3 python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | awk '{printf "%s ", $(NF-4)}'; echo -n "| "
  done
  echo -n " tridiag gen K=160/80/60: "
  for rep in 1 2; do
    GT4MI_CODEGEN_TOP_CACHE_PIPELINE=$1 GT4MI_CODEGEN_TOP_CACHE_LOOKAHEAD=$2 GT4MI_CODEGEN_TOP_CACHE_UNROLL_LDS=$3 python3 scripts/bench_generic.py --iters 20 --only tridiagonal 2>/dev/null | grep generated | awk '{printf "%s ", $(NF-4)}'; echo -n "| "
  done
  echo
done
