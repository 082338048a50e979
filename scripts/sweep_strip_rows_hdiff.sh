#!/bin/bash
# Generated horizontal diffusion (strip kernels): J rows per lane x rows per XCD run, clocks settled (200 launches), ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
for rows in 4 5 6; do
for xcd in 0 2 4; do
  echo -n "vector_rows=$rows xcd_rows=$xcd  "
  GT4MI_CODEGEN_VECTOR_ROWS=$rows GT4MI_CODEGEN_XCD_ROWS=$xcd python3 scripts/bench_generic.py --iters 200 --only horizontal_diffusion 2>/dev/null | grep -E "generated" | tr '\n' '|'
  echo
done
done
done
