#!/bin/bash
# Sweep the column-kernel knobs of the code generator on one box (tridiagonal solve + vertical advection).
out=gpurun_out/column_sweep.log; : > $out
for blk in "64,4" "64,2" "64,1" "128,2" "32,8"; do
  for pl in 24 40 64; do
    for pf in 8 4; do
      echo "== block_column=$blk prefetch=$pf prefetch_loads=$pl" >> $out
      for only in tridiag vertical; do
        GT4MI_CODEGEN_BLOCK_COLUMN=$blk GT4MI_CODEGEN_PREFETCH=$pf GT4MI_CODEGEN_PREFETCH_LOADS=$pl \
          python scripts/bench_generic.py --only $only 2>&1 | grep generated >> $out
      done
    done
  done
done
cat $out
