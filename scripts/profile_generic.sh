#!/bin/bash
# rocprofv3 kernel statistics of the generic executor's kernels next to the kernel library
# (run on the GPU box through gpurun):  scripts/profile_generic.sh <tag>
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_generic_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o gen -- python3 "$R/scripts/bench_generic.py" --iters 10 > "$OUT/stdout.log" 2>&1
f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$R/gpurun_out/generic_kernel_stats_$TAG.csv"
tail -12 "$OUT/stdout.log"
