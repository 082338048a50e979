#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + the two PMC passes for bench.py,
# then summarise into gpurun_out/.  Counters are collected in their own runs with --kernel-trace
# only, FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits; MI355X_MICROARCH.md).
#   usage: scripts/profile_bench.sh <tag> <git sha of HEAD (the box has no .git)>
set -u
TAG=${1:-r1}
SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o lap -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-other-kernels --no-allocator-variants > "$OUT/stats_stdout.log" 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o lap -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-other-kernels --no-allocator-variants > "$OUT/fetch_stdout.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o lap -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-other-kernels --no-allocator-variants > "$OUT/write_stdout.log" 2>&1
python3 "$R/scripts/summarize_profile.py" "$OUT" "$TAG" "$SHA"
