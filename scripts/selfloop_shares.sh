#!/bin/bash
# Run on the GPU box: the Laplacian's share of one rank of 8 on the 1-GPU self-loop (every neighbour the rank itself), per process
# grid: the calibrated headline and what RCCL / the direct transport each achieve.   usage: scripts/selfloop_shares.sh <tag> [grids]
set -u
TAG=${1:-r4}; GRIDS=${2:-"4x2 2x4 1x8"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
export GT4MI_BENCH_TIMESTEP=0
for G in $GRIDS; do
  python3 "$R/bench.py" --dist-selfloop --selfloop-grid $G --steps 200 --warmup 20 > "$OUT/${TAG}_bench_lap512_selfloop_share_${G}.json" 2> "$OUT/${TAG}_bench_lap512_selfloop_share_${G}.stderr"
  python3 - "$OUT/${TAG}_bench_lap512_selfloop_share_${G}.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(c["decomposition"] if "decomposition" in c else "", c["local_domain"], "headline ms/apply", d["ms_per_step"], c["schedule"], c["halo_transport"][:6], c["message_table"][:12],
      "| rccl best", d.get("rccl_best_ms_per_apply"), d.get("rccl_best_form"), "| direct best", d.get("direct_best_ms_per_apply"), d.get("direct_best_form"),
      "| kernel alone", d["roofline"]["kernel_ms"], "| candidates run/skipped", d.get("calibration_candidates_run"), d.get("calibration_candidates_skipped_for_time"))
PY
done
