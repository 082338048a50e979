out=gpurun_out/lap_matrix.log; : > $out
for rows in 4 8; do for blk in "64,4,1,1" "64,2,1,1" "64,1,1,1"; do for x in 0 4; do
  echo "== vector_rows=$rows block_ijk=$blk xcd_rows=$x" >> $out
  GT4MI_CODEGEN_VECTOR_ROWS=$rows GT4MI_CODEGEN_BLOCK_IJK=$blk GT4MI_CODEGEN_XCD_ROWS=$x python scripts/bench_generic.py --iters 60 --only laplacian 2>&1 | grep generated >> $out
done; done; done
cat $out
