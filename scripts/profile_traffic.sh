#!/bin/bash
# HBM traffic (rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE) of every kernel that
# scripts/bench_generic.py launches:  scripts/profile_traffic.sh <tag> [--only <name>]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_traffic_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o t -- python3 "$R/scripts/bench_generic.py" --iters 3 "$@" > "$OUT/$c.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c and "gt4mi" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:60]][c].append(float(row["Counter_Value"]))
print("kernel, launches, FETCH_SIZE KiB (x2 on gfx950 = bytes/512), WRITE_SIZE KiB -> GB read / GB written per launch")
for k, v in sorted(acc.items()):
    f = sorted(v["FETCH_SIZE"])[len(v["FETCH_SIZE"]) // 2] if v["FETCH_SIZE"] else float("nan")
    w = sorted(v["WRITE_SIZE"])[len(v["WRITE_SIZE"]) // 2] if v["WRITE_SIZE"] else float("nan")
    print(f"{k:60s} n={len(v['FETCH_SIZE']):2d} fetch={f:12.0f} write={w:12.0f}  read={f * 2 * 1024 / 1e9:7.3f} GB  written={w * 1024 / 1e9:7.3f} GB")
PY
