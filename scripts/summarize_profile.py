"""Summarise the rocprofv3 outputs of scripts/profile_bench.sh into two small files:

  <out>/summary_<tag>.json   kernel-stats row of the dominant kernel + PMC means
  <out>/hbm_traffic_<tag>.json   {"lap5_f64_512": bytes per launch}  (copied to profiles/hbm_traffic.json)

HBM traffic per launch = FETCH_SIZE * 2 + WRITE_SIZE (both in KiB): on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, section HBM); the factor
is verified in the same run on PyTorch's elementwise kernels whose byte counts are known.
"""

import csv
import json
import pathlib
import sys

out = pathlib.Path(sys.argv[1])
tag = sys.argv[2]
git_sha = sys.argv[3] if len(sys.argv) > 3 else "unknown"  # the GPU box has no .git: the caller passes HEAD
KERNEL = "lap5_strip_kernel"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench  # noqa: E402 - kernel_source_hash: the same function bench.py checks the committed number with


def counter_mean(path, counter, needle):
    """Mean / count / min / max of a counter over the launches of ONE kernel: of the instantiations whose name holds
    `needle`, the one with the largest mean (bench.py also launches the Laplacian on a tiny domain, many times, to
    time the host side of a call: a different template instance, which must not be averaged in)."""
    groups = {}
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"] and r["Counter_Name"] == counter:
            groups.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    if not groups:
        return None, 0, None, None
    vals = max(groups.values(), key=lambda v: sum(v) / len(v))
    return sum(vals) / len(vals), len(vals), min(vals), max(vals)


summary = {"tag": tag}
stats = out / "stats" / "lap_kernel_stats.csv"
if stats.exists():
    rows = [r for r in csv.DictReader(open(stats)) if KERNEL in r["Name"]]
    if rows:
        r = max(rows, key=lambda r: float(r["AverageNs"]))  # the 512^3 instance, not the host-cost probe's
        summary["kernel_stats"] = {"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                   "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                   "max_ns": float(r["MaxNs"]), "percentage": float(r["Percentage"])}
fetch, nf, fmin, fmax = counter_mean(out / "fetch" / "lap_counter_collection.csv", "FETCH_SIZE", KERNEL)
write, nw, wmin, wmax = counter_mean(out / "write" / "lap_counter_collection.csv", "WRITE_SIZE", KERNEL)
# calibration: a torch elementwise kernel that reads one 514x514x512 fp64 tensor (1,056,800 KiB)
cal, nc, _, _ = counter_mean(out / "fetch" / "lap_counter_collection.csv", "FETCH_SIZE", "vectorized_elementwise_kernel")
summary["pmc"] = {"FETCH_SIZE_KiB_mean": fetch, "launches": nf, "FETCH_SIZE_KiB_min": fmin, "FETCH_SIZE_KiB_max": fmax,
                  "WRITE_SIZE_KiB_mean": write, "WRITE_SIZE_KiB_min": wmin, "WRITE_SIZE_KiB_max": wmax,
                  "calibration_kernel_FETCH_SIZE_KiB": cal, "calibration_known_read_KiB": 514 * 514 * 512 * 8 / 1024}
if fetch is not None and write is not None:
    traffic = (2.0 * fetch + write) * 1024.0
    summary["hbm_bytes_per_launch"] = traffic
    summary["algorithmic_bytes_per_launch"] = 16.0 * 512**3
    summary["traffic_over_algorithmic"] = traffic / (16.0 * 512**3)
    by_name = {}
    for r in csv.DictReader(open(out / "fetch" / "lap_counter_collection.csv")):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            by_name.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    kernel_names = [max(by_name, key=lambda n: sum(by_name[n]) / len(by_name[n]))]
    (out / f"hbm_traffic_{tag}.json").write_text(json.dumps({
        "lap5_f64_512": {"bytes_per_launch": round(traffic), "kernel": kernel_names, "git_sha": git_sha,
                         "kernel_source_sha": bench.kernel_source_hash("lap5_f64_512"),
                         "source": f"profiles/{tag}_bench_lap512_summary.json"},
        "_note": "bytes per launch of lap5_strip_kernel on 512^3 fp64 = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024, "
                 "rocprofv3 --pmc, separate passes; factor 2 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts "
                 "128-B requests as 64 B), checked against a torch kernel of known size in the same run.  bench.py "
                 "reports the number only while kernel_source_sha equals the hash of the kernel sources in the tree "
                 "(bench.KERNEL_SOURCES)"}, indent=1))
    summary["git_sha"] = git_sha
    summary["kernel_source_sha"] = bench.kernel_source_hash("lap5_f64_512")
(out / f"summary_{tag}.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1))
