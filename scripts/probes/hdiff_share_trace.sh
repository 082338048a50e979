#!/bin/bash
# Run on the GPU box: BASELINE configs[4]'s share on the 1-GPU self-loop -- the calibrated line, then ONE pinned form with its kernel
# timeline.   usage: scripts/probes/hdiff_share_trace.sh [form] [transports]
FORM=${1:-fused_single_phase_inline_wg0_edge16_direct}; TR=${2:-direct}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
export GT4MI_BENCH_TIMESTEP=0
export GT4MI_BENCH_FORM=$FORM GT4MI_BENCH_TRANSPORTS=$TR
D=$OUT/hd_trace_tmp; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o hd -- python3 $R/bench.py --workload hdiff2048 --dist-selfloop --steps 40 --warmup 8 > $D.stdout 2> $D.stderr
python3 - $D.stdout <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("pinned:", d["ms_per_step"], c["apply_form"], c["calibration_ms_per_apply"], d.get("calibration_candidates_failed"))
PY
grep -i "failed\|REJECT" $D.stderr | head -3 | cut -c1-300
python3 $R/scripts/trace_timeline.py "$(find $D -name '*kernel_trace.csv' | head -1)" ring_kernel 2 | head -12
rm -rf $D $D.stdout $D.stderr
