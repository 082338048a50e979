#!/bin/bash
# Workgroup shape of the generated horizontal (strip) kernels: 64 x 4 (default) next to 128 x 2 and 256 x 1, on ONE box.
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
for rep in 1 2; do
for b in "64,4,1,1" "128,2,1,1" "256,1,1,1"; do
  for only in laplacian horizontal_diffusion; do
    echo -n "block_ijk=$b  "
    GT4MI_CODEGEN_BLOCK_IJK=$b python3 scripts/bench_generic.py --iters 200 --only $only 2>/dev/null | grep -E "generated" | tr '\n' '|'
    echo
  done
done
done
