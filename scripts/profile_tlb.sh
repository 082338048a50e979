#!/bin/bash
# Address-translation counters of the column kernels next to the Laplacian (rocprofv3 PMC, one pass per group):
#   scripts/profile_tlb.sh <tag>
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_tlb_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
             "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/g$i" -o t -- "$R/gt4py_amd/lib/microbench" tripmc > "$OUT/g$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "fill_kernel" in k or "diff_kernel" in k:
            continue
        acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k)
    for c, vals in sorted(v.items()):
        vals = sorted(vals)
        print(f"    {c:50s} n={len(vals):3d} median={vals[len(vals) // 2]:16.0f}")
PY
