#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (rocprofv3 PMC, separate passes) of every kernel a microbench section launches:
#   scripts/profile_microbench_traffic.sh <tag> <section> [more sections]
set -u
TAG=${1:-r2}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_mb_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
"$R/gt4py_amd/lib/microbench" "$@" > "$OUT/timing.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o t -- "$R/gt4py_amd/lib/microbench" "$@" > "$OUT/$c.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if row.get("Counter_Name") == c and "fill_kernel" not in k and "diff_kernel" not in k:
                acc[k.split("(")[0][:110]][c].append(float(row["Counter_Value"]))
print("kernel | launches | read GB (2 x FETCH_SIZE KiB) | written GB")
for k, v in sorted(acc.items()):
    f = sorted(v["FETCH_SIZE"])[len(v["FETCH_SIZE"]) // 2] if v["FETCH_SIZE"] else float("nan")
    w = sorted(v["WRITE_SIZE"])[len(v["WRITE_SIZE"]) // 2] if v["WRITE_SIZE"] else float("nan")
    print(f"{k:110s} n={len(v['FETCH_SIZE']):3d} read={f * 2 * 1024 / 1e9:7.4f} written={w * 1024 / 1e9:7.4f}")
PY
cat "$OUT/timing.log"
