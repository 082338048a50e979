#!/bin/bash
# Register levels of the `_tc` kernels: batches (0) next to the rolling prefetch (2), at one depth and with the default
# ladder, on ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2 3; do
for cfg in "0 104,163840" "2 104,163840" "0 -1,163840" "2 -1,163840"; do
  set -- $cfg
  echo -n "pipeline=$1 top_cache=$2  "
  GT4MI_CODEGEN_TOP_CACHE_PIPELINE=$1 GT4MI_CODEGEN_TOP_CACHE=$2 python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | grep -E "160\)" | grep generated | tr '\n' '|'
  GT4MI_CODEGEN_TOP_CACHE_PIPELINE=$1 GT4MI_CODEGEN_TOP_CACHE=$2 python3 scripts/bench_generic.py --iters 20 --only tridiagonal 2>/dev/null | grep -E "160\)" | grep generated | tr '\n' '|'; echo
done
done
