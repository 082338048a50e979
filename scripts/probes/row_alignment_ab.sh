#!/bin/bash
# Run on the GPU box: the storage preset's row alignment, 256 bytes (gt:gpu's 32 fp64 items, rounds 1-3) against 64 (round 4), through
# bench.py: the N = 1 line with its other kernels, the Laplacian's shares of 8 ranks and BASELINE configs[4]'s share on the self-loop.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for A in ${ALIGNS:-256 128 256 128}; do
  export GT4PY_AMD_ROW_ALIGN_BYTES=$A
  echo "== rows aligned to $A bytes"
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/rowalign_$A.json 2> $OUT/rowalign_$A.stderr
  python3 - $OUT/rowalign_$A.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  lap512^3", d["value"], "GLUPS frac", d["roofline"]["frac"], "|", {k.split("_")[0] + "_" + k.split("_")[-1]: v.get("glups") for k, v in d["other_kernels"].items()})
PY
  GT4MI_BENCH_TRANSPORTS=direct scripts/selfloop_shares.sh rowalign_$A "4x2 2x4 1x8" | cut -c1-60,230-262
  GT4MI_BENCH_TIMESTEP=0 GT4MI_BENCH_TRANSPORTS=direct python3 bench.py --workload hdiff2048 --dist-selfloop --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  hdiff share', d['ms_per_step'], d['config']['apply_form'], 'kernel', d['roofline']['kernel_ms'])"
done
