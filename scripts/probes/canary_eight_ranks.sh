#!/bin/bash
# Run on the GPU box: bench.py's canary command (`python -m gt4py_amd.distributed --transport direct`) as EIGHT processes on the one
# device, N times; prints the checks that failed.   usage: scripts/probes/canary_eight_ranks.sh [runs] [domain I J K]
RUNS=${1:-1}; DI=${2:-256}; DJ=${3:-192}; DK=${4:-8}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for run in $(seq 1 $RUNS); do
  D=$(mktemp -d)
  for r in 0 1 2 3 4 5 6 7; do
    GT4MI_RENDEZVOUS_FILE=$D/rdv RANK=$r WORLD_SIZE=8 LOCAL_RANK=0 PYTHONPATH=$R timeout 900 python3 -m gt4py_amd.distributed --transport direct --domain $DI $DJ $DK > $OUT/canary8_rank$r.out 2> $OUT/canary8_rank$r.err &
  done
  wait
  echo "run $run: $(tail -1 $OUT/canary8_rank0.out)"
  if ! grep -q "all correct" $OUT/canary8_rank0.out; then
    grep "WRONG" $OUT/canary8_rank0.out | cut -c1-600 | head -6
    for r in 0 1 2 3 4 5 6 7; do cp $OUT/canary8_rank$r.out $OUT/canary8_failed_run${run}_rank$r.out; grep -v "Gloo\|amdgpu.ids" $OUT/canary8_rank$r.err | tail -3 | cut -c1-300; done
  fi
done
