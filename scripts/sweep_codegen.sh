# tuning sweep of the generated kernels' launch geometry (run on the GPU box)
for cfg in "64,2,4,1" "64,2,1,2" "64,2,2,2" "64,2,4,2" "64,4,1,2" "64,4,2,2" "64,1,4,2" "64,2,1,4" "64,2,2,4" "64,4,1,4" "32,4,2,2" "32,8,1,2"; do
  echo "== block_ijk $cfg"; GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only lap 2>&1 | grep generated
  GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only horizontal_diffusion 2>&1 | grep generated
done
