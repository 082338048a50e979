cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo -n "per-stage shape (8 rows, XCD runs of 4 for light stages): "; python3 scripts/bench_generic.py --iters 30 --only laplacian 2>/dev/null | grep generated
  echo -n "round-1 shape (4 rows, no XCD grouping):                "; GT4MI_CODEGEN_VECTOR_ROWS=4 GT4MI_CODEGEN_XCD_ROWS=0 python3 scripts/bench_generic.py --iters 30 --only laplacian 2>/dev/null | grep generated
done
python3 scripts/bench_generic.py --iters 30 --only column_sum 2>/dev/null | grep generated
GT4MI_CODEGEN_VECTOR_ROWS=4 GT4MI_CODEGEN_XCD_ROWS=0 python3 scripts/bench_generic.py --iters 30 --only column_sum 2>/dev/null | grep generated
