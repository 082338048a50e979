#!/bin/bash
# Run on the GPU box: bench.py's N = 8 code path with EIGHT REAL RANKS on the one device (GT4MI_BENCH_ONE_DEVICE: gloo group, the
# direct transport over hipIpc, no RCCL) -- a rehearsal of the control flow, never a measurement.
#   usage: scripts/probes/bench_eight_ranks_one_device.sh [workload lap512|hdiff2048] [ranks]
W=${1:-lap512}; N=${2:-8}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
export GT4MI_BENCH_ONE_DEVICE=1 GT4MI_BENCH_CALIBRATION_SECONDS=20 GT4MI_BENCH_DIRECT_CALIBRATION_SECONDS=30 GT4MI_BENCH_INFORMATIONAL_SECONDS=15
export GT4MI_BENCH_DIRECT_TIMEOUT_MS=${GT4MI_BENCH_DIRECT_TIMEOUT_MS:-120000} GT4MI_BENCH_VERBOSE=1
PORT=$(python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
time timeout 1500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps 20 --warmup 3 --workload $W > $OUT/bench_one_device_${W}_$N.json 2> $OUT/bench_one_device_${W}_$N.stderr
echo "status $?"
python3 - $OUT/bench_one_device_${W}_$N.json <<'PY'
import json, sys
lines = [ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")]
print(len(lines), "JSON line(s)")
d = json.loads(lines[-1]); c = d["config"]
print({k: d.get(k) for k in ("n_gpus", "value", "ms_per_step", "rccl_nranks", "rank_devices", "rccl_matches_n_gpus", "transport_fallback", "rccl_best_ms_per_apply", "direct_best_ms_per_apply", "direct_best_form", "calibration_candidates_run", "calibration_candidates_skipped_for_time", "calibration_candidates_failed", "provisional", "deadline_exceeded")})
print({k: c.get(k) for k in ("decomposition", "local_domain", "transport", "halo_transport", "schedule", "message_table", "apply_form", "direct_transport_canary", "direct_transport_dropped_at")}, c.get("verified", {}).get("headline_form_correct_on_every_rank"), c.get("verified", {}).get("forms_checked"), c.get("verified", {}).get("forms_rejected"))
print("calibration:", c.get("calibration_ms_per_apply"))
print("extra keys:", list((d.get("extra") or {}).keys()))
PY
grep -v "Gloo\|amdgpu.ids\|^$" $OUT/bench_one_device_${W}_$N.stderr | tail -12 | cut -c1-300
