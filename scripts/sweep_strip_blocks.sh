#!/bin/bash
# Generated strip kernels (`_vec`: Laplacian; `_vecs`: horizontal diffusion): workgroup shape x rows per lane x XCD runs,
# GLUPS through scripts/bench_generic.py on ONE box (round 3, VERDICT item 6: the hand-written Laplacian tiles 512 columns
# x 8 rows per workgroup, the generated one 128 x 32).
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GT4PY_AMD_CACHE_DIR=""
glups() { python3 scripts/bench_generic.py --iters 100 --only "$1" 2>/dev/null | grep generated | grep -v "_if\|_f32" | awk '{printf "%s ", $(NF-4)}'; }
echo "== generated Laplacian 512^3 (hand-written: $(python3 scripts/bench_generic.py --iters 100 --only laplacian 2>/dev/null | grep library | awk '{print $(NF-4)}') GLUPS)"
for blk in "64,4,1,1" "128,2,1,1" "256,1,1,1" "128,1,1,1"; do
  for rows in 8; do
    for x in 0 2 4 8 16; do
      echo -n "block=$blk rows=$rows xcd_rows=$x  "; GT4MI_CODEGEN_BLOCK_IJK=$blk GT4MI_CODEGEN_VECTOR_ROWS=$rows GT4MI_CODEGEN_XCD_ROWS=$x glups laplacian; echo
    done
  done
done
echo "== generated horizontal diffusion fp64 512x1024x80 (hand-written: $(python3 scripts/bench_generic.py --iters 100 --only horizontal_diffusion 2>/dev/null | grep library | grep -v f32 | head -1 | awk '{print $(NF-4)}') GLUPS)"
for blk in "64,4,1,1" "64,2,1,1" "64,8,1,1" "128,2,1,1"; do
  for rows in 4 8; do
    for x in 0 2 4 8; do
      echo -n "block=$blk rows=$rows xcd_rows=$x  "; GT4MI_CODEGEN_BLOCK_IJK=$blk GT4MI_CODEGEN_SHARED_ROWS=$rows GT4MI_CODEGEN_SHARED_XCD_ROWS=$x glups horizontal_diffusion; echo
    done
  done
done
