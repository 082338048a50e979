# tuning sweep of the generated kernels (run on the GPU box); results: profiles/r1_codegen_sweep.log
for rows in 4 8; do for cfg in "64,4,1,1" "128,2,1,1" "256,1,1,1" "128,4,1,1" "256,2,1,1" "128,1,1,1" "512,1,1,1"; do
  echo "== vector_rows $rows block_ijk $cfg"
  GT4MI_CODEGEN_VECTOR_ROWS=$rows GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only lap 2>&1 | grep generated
  GT4MI_CODEGEN_VECTOR_ROWS=$rows GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only horizontal_diffusion 2>&1 | grep generated
done; done
