# tuning sweep of the generated kernels (run on the GPU box); results: profiles/r1_codegen_sweep.log
for nt in 0 1; do for r in 0 4; do for cfg in "64,2,4,1" "64,2,2,2" "64,4,1,1" "256,1,1,8"; do
  echo "== nontemporal $nt xcd_rows $r block_ijk $cfg"
  GT4MI_CODEGEN_NONTEMPORAL=$nt GT4MI_CODEGEN_XCD_ROWS=$r GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only lap 2>&1 | grep generated
  GT4MI_CODEGEN_NONTEMPORAL=$nt GT4MI_CODEGEN_XCD_ROWS=$r GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only horizontal_diffusion 2>&1 | grep generated
done; done; done
for u in 1 4 8; do for b in "64,4" "128,1"; do
  echo "== unroll $u block_column $b"
  GT4MI_CODEGEN_UNROLL=$u GT4MI_CODEGEN_BLOCK_COLUMN=$b python scripts/bench_generic.py --only vertical 2>&1 | grep generated
  GT4MI_CODEGEN_UNROLL=$u GT4MI_CODEGEN_BLOCK_COLUMN=$b python scripts/bench_generic.py --only tridiag 2>&1 | grep generated
done; done
