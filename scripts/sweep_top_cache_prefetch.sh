cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for pl in 40 64 96; do
  echo -n "prefetch_loads=$pl  "
  GT4MI_CODEGEN_PREFETCH_LOADS=$pl python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | grep -E "generated" | tr '\n' '|'; echo
done
for bc in "64,1" "64,2" "64,4"; do
  echo -n "block_column=$bc  "
  GT4MI_CODEGEN_BLOCK_COLUMN=$bc python3 scripts/bench_generic.py --iters 20 --only vertical_advection 2>/dev/null | grep -E "generated" | tr '\n' '|'; echo
done
done
