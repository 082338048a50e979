#!/bin/bash
# Run on the GPU box: the tridiagonal solve under the four workgroup -> (I tile, J row) mappings of tridiag_pipe_kernel (MAP 0-3):
# launch times (microbench trimap) and the address-translation counters per mapping (rocprofv3 --pmc, one pass per group, the
# program directly after `--`).   usage: scripts/profile_tridiag_map.sh <tag>   ->  gpurun_out/<tag>_tridiag_translation.txt
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_trimap_$TAG
LOG=$R/gpurun_out/${TAG}_tridiag_translation.txt
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
"$R/gt4py_amd/lib/microbench" trimap > "$OUT/times.log" 2>&1
{ echo "== launch times (microbench trimap; fp64 1024x1024x160, two passes; then 200x301x150) =="; grep -E "tridiag64|check|device=" "$OUT/times.log"; } > "$LOG"
i=0
for group in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
             "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/g$i" -o t -- "$R/gt4py_amd/lib/microbench" trimap > "$OUT/g$i.log" 2>&1
done
python3 - "$OUT" >> "$LOG" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "tridiag_pipe_kernel" not in k or int(row.get("Grid_Size", "0") or 0) < 1024 * 1024:  # the 1024x1024x160 launches only
            continue
        acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== translation counters per launch, median over the launches of the 1024x1024x160 solve (template arguments: T, register levels,")
print("   LDS levels, batch, waves per workgroup, MAP) ==")
for k, v in sorted(acc.items()):
    print(k)
    for c, vals in sorted(v.items()):
        vals = sorted(vals)
        print(f"    {c:50s} n={len(vals):3d} median={vals[len(vals) // 2]:16.0f}")
    if "GRBM_UTCL2_BUSY" in v and "GRBM_GUI_ACTIVE" in v:
        b, a = sorted(v["GRBM_UTCL2_BUSY"]), sorted(v["GRBM_GUI_ACTIVE"])
        print(f"    {'UTCL2 busy / GUI active':50s}       {b[len(b) // 2] / a[len(a) // 2]:16.3f}")
PY
rm -rf "$OUT"/g*/
cat "$LOG"
