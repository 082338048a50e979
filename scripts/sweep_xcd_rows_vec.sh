#!/bin/bash
# XCD-aware tile order for the generated 16-byte-lane strip kernels (halo rows shared by vertically adjacent strips).
out=gpurun_out/xcd_rows_vec.log; : > $out
for rep in 1 2; do
for r in 0 1 2 4 8 16; do
  echo "== rep $rep xcd_rows=$r" >> $out
  for only in laplacian horizontal_diffusion; do
    GT4MI_CODEGEN_XCD_ROWS=$r python scripts/bench_generic.py --iters 50 --only $only 2>&1 | grep generated >> $out
  done
done
done
cat $out
