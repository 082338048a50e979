#!/bin/bash
# `_vecs` strip kernels: rows per lane x XCD runs, fp64 / fp32 horizontal diffusion (GLUPS), ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
for rows in 4 5 6 7 8; do
for x in 2 4 8; do
  echo -n "rows=$rows xcd_rows=$x  "; GT4MI_CODEGEN_SHARED_ROWS=$rows GT4MI_CODEGEN_SHARED_XCD_ROWS=$x python3 scripts/bench_generic.py --iters 200 --only horizontal_diffusion 2>/dev/null | grep generated | grep -v "_if" | awk '{printf "%s ", $(NF-4)}'; echo
done
done
done
