#!/bin/bash
# Strip kernel with shared temporaries (`_vecs`): J rows per lane, next to the recomputing strip kernel, ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
echo -n "recomputing (_vec)  "; GT4MI_CODEGEN_SHARED_TEMPORARIES=0 python3 scripts/bench_generic.py --iters 200 --only horizontal 2>/dev/null | grep generated | awk '{printf "%s ", $(NF-4)}'; echo
for rows in 2 3 4 5 6 8; do
  echo -n "shared rows=$rows  "; GT4MI_CODEGEN_SHARED_ROWS=$rows python3 scripts/bench_generic.py --iters 200 --only horizontal 2>/dev/null | grep generated | awk '{printf "%s ", $(NF-4)}'; echo
done
done
