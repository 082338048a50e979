#!/bin/bash
# Workgroup shape of the generated column kernels with the deep top-of-column cache: 1, 2 or 4 waves per workgroup,
# always 4 waves per CU (40 LDS levels per column), on ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
for cfg in "64,4 104,163840" "64,2 104,81920" "64,1 104,40960" "64,1 112,40960" "64,1 16,40960" "64,4 0,0" "64,1 0,0" "64,2 0,0"; do
  set -- $cfg
  echo -n "block_column=$1 top_cache=$2  "
  GT4MI_CODEGEN_BLOCK_COLUMN=$1 GT4MI_CODEGEN_TOP_CACHE=$2 python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | grep -E "generated" | tr '\n' '|'
  GT4MI_CODEGEN_BLOCK_COLUMN=$1 GT4MI_CODEGEN_TOP_CACHE=$2 python3 scripts/bench_generic.py --iters 20 --only tridiagonal 2>/dev/null | grep -E "generated" | tr '\n' '|'; echo
done
done
