#!/bin/bash
# The same tridiagonal solve in N fresh processes on ONE box: speed next to the virtual addresses of its five arrays.
# (Are the fast / slow modes a property of the addresses a process happens to get?)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in $(seq 1 ${1:-10}); do
  GT4MI_PRINT_PTRS=1 python3 scripts/bench_generic.py --iters 20 --only tridiagonal 2>/dev/null | grep -E "160\)|full" | head -3 | awk '/full/{sub(/.*full:/,""); a=$0} /library/{printf "%s GLUPS library ", $(NF-4)} /generated/{printf "%s generated ", $(NF-4)} END{print a}'
done
