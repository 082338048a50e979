#!/bin/bash
# fewer, deeper columns per CU: block shapes of the column kernels x LDS depth cap of the top-of-column cache
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for cfg in "64,4 16,163840,64" "64,2 16,163840,128" "64,2 16,81920,128" "64,3 16,163840,128" "64,1 16,81920,160" "64,8 16,163840,64"; do
  set -- $cfg
  echo -n "block_column=$1 top_cache=$2  "
  GT4MI_CODEGEN_BLOCK_COLUMN=$1 GT4MI_CODEGEN_TOP_CACHE=$2 python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | grep -E "generated" | tr '\n' '|'
  GT4MI_CODEGEN_BLOCK_COLUMN=$1 GT4MI_CODEGEN_TOP_CACHE=$2 python3 scripts/bench_generic.py --iters 20 --only tridiagonal 2>/dev/null | grep -E "generated" | tr '\n' '|'; echo
done
done
