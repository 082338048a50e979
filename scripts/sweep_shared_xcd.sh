#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
for x in 0 1 2 4 8; do
  echo -n "shared rows=5 xcd_rows=$x  "; GT4MI_CODEGEN_SHARED_XCD_ROWS=$x python3 scripts/bench_generic.py --iters 200 --only horizontal 2>/dev/null | grep generated | awk '{printf "%s ", $(NF-4)}'; echo
done
done
