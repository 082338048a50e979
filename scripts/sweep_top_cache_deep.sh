#!/bin/bash
# Generated two-sweep column kernels with DEEP register caches (possible since the register levels are pinned and the
# second sweep has its own bases, hip_codegen._pin_register_level / _second_sweep_bases), on ONE box:
#   scripts/sweep_top_cache_deep.sh  ->  lines of scripts/bench_generic.py per (register levels, LDS bytes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
for tc in "16,163840" "48,163840" "80,163840" "96,163840" "104,163840" "112,163840" "120,163840" "-1,163840"; do
  for only in vertical_advection tridiagonal; do
    echo -n "top_cache=$tc  "
    GT4MI_CODEGEN_TOP_CACHE=$tc python3 scripts/bench_generic.py --iters 20 --only $only 2>/dev/null | grep -E "generated|library" | tr '\n' '|'
    echo
  done
done
done
