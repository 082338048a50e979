#!/bin/bash
# Run on the GPU box: kernel traces (rocprofv3 --kernel-trace) of the fused distributed hdiff step on the 1-GPU self-loop for
# the two schedules x occupancy throttles of the interior kernel; prints mean kernel durations and two step timelines each.
#   usage: scripts/dist_trace_sweep.sh <tag> "<schedule:wg_per_cu> ..."
set -u
TAG=${1:-r3}
CASES=${2:-"0:0 0:3 1:0 1:3 1:2"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
LOG=$OUT/${TAG}_dist_trace_sweep.txt
: > "$LOG"
cd /tmp && export TMPDIR=/tmp
export GT4MI_BENCH_FORM=${GT4MI_BENCH_FORM:-fused_single_phase}
for C in $CASES; do
  export GT4MI_DIST_SCHEDULE=${C%%:*} GT4MI_DIST_INTERIOR_WG_PER_CU=${C##*:}
  D=$OUT/${TAG}_trace_tmp
  rm -rf "$D"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$D" -o hd -- python3 "$R/bench.py" --workload hdiff2048 --dist-selfloop --steps 30 --warmup 5 > "$D.stdout" 2>"$D.stderr"
  echo "===== schedule=$GT4MI_DIST_SCHEDULE interior_wg_per_cu=$GT4MI_DIST_INTERIOR_WG_PER_CU form=$GT4MI_BENCH_FORM: $(python3 -c "import json,sys; d=json.loads(open('$D.stdout').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'calib', d['config']['calibration_ms_per_apply'])" 2>&1)" >> "$LOG"
  python3 "$R/scripts/trace_timeline.py" "$(find $D -name '*kernel_trace.csv' | head -1)" ring_kernel 2 >> "$LOG" 2>&1
  rm -rf "$D" "$D.stdout" "$D.stderr"
done
cat "$LOG"
