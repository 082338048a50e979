#!/bin/bash
# Generated two-sweep column kernels with different depths of the top-of-column cache, on ONE box:
#   scripts/sweep_top_cache.sh  ->  lines of scripts/bench_generic.py per (register levels, LDS bytes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2; do
for tc in "0,0" "16,163840" "8,163840" "24,163840" "0,163840"; do
  for only in vertical_advection tridiagonal; do
    echo -n "top_cache=$tc  "
    GT4MI_CODEGEN_TOP_CACHE=$tc python3 scripts/bench_generic.py --iters 20 --only $only 2>/dev/null | grep -E "generated|library" | tr '\n' '|'
    echo
  done
done
done
