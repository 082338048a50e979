#!/bin/bash
# Run on the GPU box (through gpurun): the 1-GPU self-loop rehearsal of the distributed steps under a sweep of
#   * the occupancy throttle of the interior kernel (GT4MI_DIST_INTERIOR_WG_PER_CU, common.hip.h: launch_dynamic_lds),
#   * RCCL's point-to-point channel count (NCCL_MIN_P2P_NCHANNELS / NCCL_MAX_P2P_NCHANNELS),
# plus a kernel trace of one fused step.   usage: scripts/dist_selfloop_sweep.sh <tag>
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
LOG=$OUT/${TAG}_dist_selfloop_sweep.log
: > "$LOG"
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s ms_per_step %.5f  kernel_ms %.5f  form %s  calib %s  extra %s' % (sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('apply_form', d['config'].get('message_table','')), d['config'].get('calibration_ms_per_apply'), (d.get('extra') or {}).get('timestep_ms_per_step')))" "$1"; }
cd "$R"
for WG in 0 4 3 2 1; do
  for CH in default 8 32; do
    if [ "$CH" = default ]; then unset NCCL_MIN_P2P_NCHANNELS NCCL_MAX_P2P_NCHANNELS; else export NCCL_MIN_P2P_NCHANNELS=$CH NCCL_MAX_P2P_NCHANNELS=$CH; fi
    export GT4MI_DIST_INTERIOR_WG_PER_CU=$WG
    timeout 200 python3 bench.py --workload hdiff2048 --dist-selfloop --steps 100 --warmup 20 2>/dev/null | line "hdiff2048 wg_per_cu=$WG p2p_channels=$CH" >> "$LOG"
  done
done
unset NCCL_MIN_P2P_NCHANNELS NCCL_MAX_P2P_NCHANNELS
for WG in 0 4 2 1; do
  export GT4MI_DIST_INTERIOR_WG_PER_CU=$WG
  timeout 300 python3 bench.py --dist-selfloop --selfloop-ranks 8 --steps 200 --warmup 20 2>/dev/null | line "lap512 share of 8 wg_per_cu=$WG" >> "$LOG"
done
unset GT4MI_DIST_INTERIOR_WG_PER_CU
cd /tmp && export TMPDIR=/tmp
export GT4MI_BENCH_FORM=fused_single_phase
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_trace_hd" -o hd -- python3 "$R/bench.py" --workload hdiff2048 --dist-selfloop --steps 20 --warmup 5 > "$OUT/${TAG}_trace_hd_stdout.log" 2>&1
python3 "$R/scripts/trace_timeline.py" "$(ls $OUT/${TAG}_trace_hd/*/*kernel_trace.csv $OUT/${TAG}_trace_hd/*kernel_trace.csv 2>/dev/null | head -1)" ring_kernel 2 > "$OUT/${TAG}_trace_hd_timeline.txt" 2>&1
rm -rf "$OUT/${TAG}_trace_hd"
cat "$LOG"; cat "$OUT/${TAG}_trace_hd_timeline.txt"
