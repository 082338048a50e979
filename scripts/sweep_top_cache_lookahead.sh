#!/bin/bash
# Rolling prefetch of the register levels: how many levels ahead, at 104 and 112 register levels, on ONE box
# (three processes per setting: the placement of a process's arrays moves these kernels by several per cent).
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for tc in 104,163840 112,163840; do
for la in 2 3 4 6 8; do
  echo -n "top_cache=$tc lookahead=$la  "
  for rep in 1 2 3; do
    GT4MI_CODEGEN_TOP_CACHE_LOOKAHEAD=$la GT4MI_CODEGEN_TOP_CACHE_PIPELINE=2 GT4MI_CODEGEN_TOP_CACHE=$tc python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | grep -E "160\)" | awk '{printf "%s ", $(NF-4)}'
  done
  echo -n " | tridiag: "
  for rep in 1 2 3; do
    GT4MI_CODEGEN_TOP_CACHE_LOOKAHEAD=$la GT4MI_CODEGEN_TOP_CACHE_PIPELINE=2 GT4MI_CODEGEN_TOP_CACHE=$tc python3 scripts/bench_generic.py --iters 20 --only tridiagonal 2>/dev/null | grep -E "160\)" | grep generated | awk '{printf "%s ", $(NF-4)}'
  done
  echo
done
done
