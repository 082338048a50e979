#!/bin/bash
# Run on the GPU box: kernel trace of the fused Laplacian apply (exchange on every apply) on the share of one rank of a PI x PJ
# grid, 1-GPU self-loop.   usage: scripts/dist_trace_lap_apply.sh <tag> <PIxPJ> <schedule join|chain|swap|swap-packed|inline> <wg_per_cu>
#                                  [single 0|1] [last kernel of a step: ring_kernel]
# GT4MI_BENCH_TRANSPORTS=direct selects the direct transport; its inline schedule is ONE launch: "lap5_step_kernel".
set -u
TAG=${1:-r3}; G=${2:-4x2}; SCHED=${3:-join}; WG=${4:-0}; SP=${5:-1}; LAST=${6:-ring_kernel}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GT4MI_BENCH_TIMESTEP=0 GT4MI_BENCH_GRID=1x1 GT4MI_BENCH_SINGLE_PHASE=$SP GT4MI_BENCH_SCHEDULE=$SCHED GT4MI_BENCH_WG_PER_CU=$WG
D=$OUT/${TAG}_trace_tmp
rm -rf "$D"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$D" -o lap -- python3 "$R/bench.py" --dist-selfloop --selfloop-grid $G --steps 40 --warmup 8 > "$D.stdout" 2>"$D.stderr"
LOG=$OUT/${TAG}_dist_trace_lap_apply_${G}_${SCHED}_wg${WG}.txt
python3 -c "import json; d=json.loads(open('$D.stdout').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['config']['local_domain'], d['config']['message_table'], d['config']['schedule'])" > "$LOG" 2>&1
python3 "$R/scripts/trace_timeline.py" "$(find $D -name '*kernel_trace.csv' | head -1)" "$LAST" 2 >> "$LOG" 2>&1
rm -rf "$D" "$D.stdout" "$D.stderr"
cat "$LOG"
