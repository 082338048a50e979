for cfg in "64,4,1" "64,4,2" "64,4,4" "128,2,1" "128,2,2" "256,1,1" "256,1,2" "256,1,4" "64,8,1" "64,2,4"; do
  echo "== block_ijk $cfg"; GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only lap 2>&1 | grep generated
  GT4MI_CODEGEN_BLOCK_IJK=$cfg python scripts/bench_generic.py --only horizontal_diffusion 2>&1 | grep generated
done
for u in 1 2 4 8; do for b in "64,4" "64,2" "64,1" "128,1"; do
  echo "== unroll $u block_column $b"; GT4MI_CODEGEN_UNROLL=$u GT4MI_CODEGEN_BLOCK_COLUMN=$b python scripts/bench_generic.py --only vertical 2>&1 | grep generated
  GT4MI_CODEGEN_UNROLL=$u GT4MI_CODEGEN_BLOCK_COLUMN=$b python scripts/bench_generic.py --only tridiag 2>&1 | grep generated
done; done
