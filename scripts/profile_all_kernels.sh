#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs: TCC
# slot limits, MI355X_MICROARCH.md) for EVERY kernel of BASELINE.json at its own size, on the tree as it is -- VERDICT round 4,
# item 3: the evidence of the non-headline kernels predated the storage change.  The program stands directly behind `--`.
#   usage: scripts/profile_all_kernels.sh <tag> <git sha of HEAD (the box has no .git)>
# Two groups, each in its own processes: the two Laplacian workloads run the same kernel instantiation and can only be told apart
# by the process they ran in.  Summary: gpurun_out/prof_all_<tag>/all_kernels_summary.json (-> profiles/<tag>_all_kernels_summary.json)
# and hbm_traffic.json (-> profiles/hbm_traffic.json, which bench.py reads under the kernel-source hash rule).
set -u
TAG=${1:-r5}
SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_all_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G1=lap5_f64_512,hdiff_limiter_f32_1024x1024x80,hdiff_limiter_f32_literal32_1024x1024x80,hdiff_limiter_f64_512x1024x80,tridiagonal_f64_1024x1024x160,generated_vertical_advection_f64_1024x1024x160,generated_laplacian_f64_512x512x512,generated_hdiff_limiter_f64_512x1024x80
G2=laplacian_f64_512x512x128_config1
for g in 1 2; do
  eval only=\$G$g
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/g$g/stats" -o k -- python3 "$R/scripts/run_baseline_kernels.py" --only "$only" --steps 50 > "$OUT/g${g}_stats_stdout.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/g$g/fetch" -o k -- python3 "$R/scripts/run_baseline_kernels.py" --only "$only" --steps 10 > "$OUT/g${g}_fetch_stdout.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/g$g/write" -o k -- python3 "$R/scripts/run_baseline_kernels.py" --only "$only" --steps 10 > "$OUT/g${g}_write_stdout.log" 2>&1
done
python3 "$R/scripts/summarize_all_kernels.py" "$OUT" "$TAG" "$SHA"
