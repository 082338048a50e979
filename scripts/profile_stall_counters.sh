#!/bin/bash
# Run on the GPU box (through gpurun): what do the horizontal-diffusion kernels wait on?  rocprofv3 --pmc passes (counters
# in runs of their own, with --kernel-trace only; the program directly after `--`) over scripts/stall_probe.py, one pass per
# group of counters that fits the block's slots (MI355X_MICROARCH.md: SQ 8, TCC 4), only counters `rocprofv3 -L` lists.
#   usage: scripts/profile_stall_counters.sh <tag>       -> gpurun_out/<tag>_hdiff_stall_counters.txt
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}_stall
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/available.txt" 2>&1
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
 "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LEVEL_WAVES"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
 "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"
 "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUSY_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
n=0
for P in "${PASSES[@]}"; do
  KEEP=""
  for C in $P; do
    if grep -qw "$C" "$OUT/available.txt"; then KEEP="$KEEP $C"; else echo "not available: $C" >> "$OUT/skipped.txt"; fi
  done
  [ -z "$KEEP" ] && continue
  n=$((n + 1))
  timeout 300 rocprofv3 --pmc $KEEP --kernel-trace --output-format csv -d "$OUT/pass$n" -o p -- python3 "$R/scripts/stall_probe.py" 6 > "$OUT/pass$n.log" 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python3 "$R/scripts/stall_probe.py" 20 > "$OUT/stats.log" 2>&1
python3 - "$OUT" > "$R/gpurun_out/${TAG}_hdiff_stall_counters.txt" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "hdiff_jmarch" in k or "lap5_strip" in k:
            tag = ("hdiff f64 512x1024x80" if "IdddL" in k or "<double" in k else "hdiff f32 1024x1024x80") if "hdiff" in k else "lap5 f64 512^3 (control)"
            acc[row["Counter_Name"]][tag].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/stats/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "hdiff_jmarch" in k or "lap5_strip" in k:
            tag = ("hdiff f64 512x1024x80" if "IdddL" in k or "<double" in k else "hdiff f32 1024x1024x80") if "hdiff" in k else "lap5 f64 512^3 (control)"
            dur[tag].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
tags = ["hdiff f64 512x1024x80", "hdiff f32 1024x1024x80", "lap5 f64 512^3 (control)"]
print("rocprofv3 --pmc, median per launch over the launches of scripts/stall_probe.py (first launch of each kernel dropped)")
print(f"{'counter':42s}" + "".join(f"{t:>28s}" for t in tags))
med = lambda v: sorted(v[1:] or v)[len(v[1:] or v) // 2]
print(f"{'kernel duration, unprofiled run (us)':42s}" + "".join(f"{med(dur[t]):28.1f}" if dur[t] else f"{'-':>28s}" for t in tags))
for c in sorted(acc):
    print(f"{c:42s}" + "".join(f"{med(acc[c][t]):28.4g}" if acc[c][t] else f"{'-':>28s}" for t in tags))
try:
    print("\nnot available on this rocprofv3:", " ".join(l.split(": ")[1].strip() for l in open(f"{out}/skipped.txt")))
except Exception:
    pass
PY
cat "$R/gpurun_out/${TAG}_hdiff_stall_counters.txt"
rm -rf "$OUT"/pass*/ "$OUT/stats"
