#!/bin/bash
# Register levels of the generated `_tc` kernels with and without the loads of the next batch in flight, on ONE box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
for cfg in "0 80,163840" "1 80,163840" "0 96,163840" "1 96,163840" "0 104,163840" "1 104,163840" "0 112,163840"; do
  set -- $cfg
  for only in vertical_advection tridiagonal; do
    echo -n "pipeline=$1 top_cache=$2  "
    GT4MI_CODEGEN_TOP_CACHE_PIPELINE=$1 GT4MI_CODEGEN_TOP_CACHE=$2 python3 scripts/bench_generic.py --iters 20 --only $only 2>/dev/null | grep -E "generated" | tr '\n' '|'
    echo
  done
done
done
