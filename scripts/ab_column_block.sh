out=gpurun_out/column_ab.log; : > $out
for rep in 1 2 3; do
for blk in "64,4" "64,2" "128,2" "64,1"; do
  echo "== rep $rep block_column=$blk" >> $out
  for only in tridiag vertical column_sum; do
    GT4MI_CODEGEN_BLOCK_COLUMN=$blk python scripts/bench_generic.py --iters 100 --only $only 2>&1 | grep generated >> $out
  done
done
done
cat $out
