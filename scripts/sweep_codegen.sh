# tuning sweep of the generated kernels (run on the GPU box); results: profiles/r1_codegen_sweep.log
for x in 0 1 2 4 8; do for rows in 4 6; do
  echo "== xcd_rows $x vector_rows $rows"
  GT4MI_CODEGEN_XCD_ROWS=$x GT4MI_CODEGEN_VECTOR_ROWS=$rows python scripts/bench_generic.py --only lap 2>&1 | grep generated
  GT4MI_CODEGEN_XCD_ROWS=$x GT4MI_CODEGEN_VECTOR_ROWS=$rows python scripts/bench_generic.py --only horizontal_diffusion 2>&1 | grep generated
done; done
