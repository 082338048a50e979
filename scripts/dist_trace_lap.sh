#!/bin/bash
# Run on the GPU box: kernel trace of the time-skewed Laplacian stepper on the share of 8 ranks (512 x 64 x 512, 1-GPU
# self-loop), ghost depth $2 (default 2); prints the timeline of two cycles.   usage: scripts/dist_trace_lap.sh <tag> [halo] [wg_per_cu]
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GT4MI_BENCH_MODE=timestep GT4MI_BENCH_HALO=${2:-2} GT4MI_BENCH_TIMESTEP=0 GT4MI_DIST_INTERIOR_WG_PER_CU=${3:-0}
D=$OUT/${TAG}_trace_tmp
rm -rf "$D"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$D" -o lap -- python3 "$R/bench.py" --dist-selfloop --selfloop-ranks 8 --steps 40 --warmup 8 > "$D.stdout" 2>"$D.stderr"
LOG=$OUT/${TAG}_dist_trace_lap_halo${GT4MI_BENCH_HALO}_wg${GT4MI_DIST_INTERIOR_WG_PER_CU}.txt
python3 -c "import json; d=json.loads(open('$D.stdout').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['config']['workload'][-120:])" > "$LOG" 2>&1
# a cycle ends with its last interior kernel; anchor on the pack kernel (one per cycle) instead
python3 "$R/scripts/trace_timeline.py" "$(find $D -name '*kernel_trace.csv' | head -1)" "halo_batch_kernel<unsigned long, true>" 3 >> "$LOG" 2>&1
rm -rf "$D" "$D.stdout" "$D.stderr"
cat "$LOG"
