#!/bin/bash
# `_tc` kernels with and without streaming stores, next to the hand-written kernels of the micro-benchmark, on ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2 3; do
for st in 0 1; do
  echo -n "streaming=$st  "
  GT4MI_CODEGEN_TOP_CACHE_STREAMING=$st python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | grep -E "generated" | tr '\n' '|'
  GT4MI_CODEGEN_TOP_CACHE_STREAMING=$st python3 scripts/bench_generic.py --iters 20 --only tridiagonal 2>/dev/null | grep -E "generated|library" | tr '\n' '|'; echo
done
done
timeout 200 gt4py_amd/lib/microbench vadv 2>&1 | grep -E "regs (16|104|112) lds 40 batch 4 " | head -4
