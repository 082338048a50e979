"""Summarise the rocprofv3 outputs of scripts/profile_all_kernels.sh:

  <out>/all_kernels_summary.json   workload -> {kernel, calls, average_ns (--kernel-trace --stats), FETCH_SIZE / WRITE_SIZE means,
                                   hbm_bytes_per_launch, algorithmic bytes, their ratio, what bench.py's HIP events measured in the
                                   same processes, kernel_source_sha, git_sha}
  <out>/hbm_traffic.json           workload -> {bytes_per_launch, kernel, git_sha, kernel_source_sha, source}: the file bench.py reads
                                   (`_committed_traffic`: a number is reported only while the hash of the kernel's sources matches)

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
read (MI355X_MICROARCH.md, section HBM: 128-B requests tallied at 64 B); Infinity-Cache hits are counted, not excluded.  The first
launch of every kernel in a process is dropped (cold caches, lazily mapped pages).
"""

import csv
import glob
import json
import pathlib
import sys

out = pathlib.Path(sys.argv[1])
tag = sys.argv[2]
git_sha = sys.argv[3] if len(sys.argv) > 3 else "unknown"
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

GROUPS = {1: [w for w in bench.KERNEL_SOURCES if w != "laplacian_f64_512x512x128_config1"], 2: ["laplacian_f64_512x512x128_config1"]}
ALGORITHMIC = {  # bytes per launch: SURVEY.md section 8d's per-update figure x the lattice updates of one launch
    "lap5_f64_512": 16.0 * 512**3, "laplacian_f64_512x512x128_config1": 16.0 * 512 * 512 * 128,
    "hdiff_limiter_f32_1024x1024x80": 12.0 * 1024 * 1024 * 80, "hdiff_limiter_f32_literal32_1024x1024x80": 12.0 * 1024 * 1024 * 80,
    "hdiff_limiter_f64_512x1024x80": 24.0 * 512 * 1024 * 80, "tridiagonal_f64_1024x1024x160": 56.0 * 1024 * 1024 * 160,
    "generated_vertical_advection_f64_1024x1024x160": 48.0 * 1024 * 1024 * 160, "generated_laplacian_f64_512x512x512": 16.0 * 512**3,
    "generated_hdiff_limiter_f64_512x1024x80": 24.0 * 512 * 1024 * 80,
}


def one(pattern):
    files = glob.glob(str(pattern), recursive=True)
    return files[0] if files else None


def counter_by_kernel(path, counter):
    groups = {}
    if path is None:
        return groups
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            groups.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return groups


def bench_numbers(log):
    try:
        line = [ln for ln in open(log) if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception:
        return {}


summary, traffic_file = {}, {}
for g, workloads in GROUPS.items():
    base = out / f"g{g}"
    stats = one(base / "stats" / "**" / "*kernel_stats.csv")
    rows = list(csv.DictReader(open(stats))) if stats else []
    fetch = counter_by_kernel(one(base / "fetch" / "**" / "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_by_kernel(one(base / "write" / "**" / "*counter_collection.csv"), "WRITE_SIZE")
    measured = bench_numbers(out / f"g{g}_stats_stdout.log")
    for w in workloads:
        needle = bench.KERNEL_NEEDLES[w]
        mine = [r for r in rows if needle in r["Name"]]
        if not mine:
            summary[w] = {"error": f"no kernel matching {needle!r} in {stats}"}
            continue
        r = max(mine, key=lambda r: float(r["AverageNs"]))  # (the instantiation that does the work, should helpers share the needle)
        name = r["Name"]
        entry = {"kernel": name.split("(")[0], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                 "max_ns": float(r["MaxNs"]), "bench_hip_events": measured.get(w)}
        f = [v for k, v in fetch.items() if k.split("(")[0] == name.split("(")[0]]
        wv = [v for k, v in write.items() if k.split("(")[0] == name.split("(")[0]]
        if f and wv:
            fvals, wvals = f[0][1:] or f[0], wv[0][1:] or wv[0]  # (first launch dropped)
            fk, wk = sum(fvals) / len(fvals), sum(wvals) / len(wvals)
            bytes_per_launch = (2.0 * fk + wk) * 1024.0
            alg = ALGORITHMIC[w]
            entry.update({"FETCH_SIZE_KiB_mean": fk, "FETCH_SIZE_launches": len(fvals), "WRITE_SIZE_KiB_mean": wk, "WRITE_SIZE_launches": len(wvals),
                          "hbm_bytes_per_launch": round(bytes_per_launch), "algorithmic_bytes_per_launch": alg,
                          "traffic_over_algorithmic": round(bytes_per_launch / alg, 4),
                          "achieved_algorithmic_gbs": round(alg / float(r["AverageNs"]), 1),
                          "frac_of_hbm_peak": round(alg / float(r["AverageNs"]) / bench.PEAK_GBS, 4)})
            traffic_file[w] = {"bytes_per_launch": round(bytes_per_launch), "kernel": [entry["kernel"]], "git_sha": git_sha,
                               "kernel_source_sha": bench.kernel_source_hash(w), "source": f"profiles/{tag}_all_kernels_summary.json"}
        entry["git_sha"], entry["kernel_source_sha"] = git_sha, bench.kernel_source_hash(w)
        summary[w] = entry
traffic_file["_note"] = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024, rocprofv3 --pmc, separate passes, every kernel of BASELINE.json "
                         "at its own size (scripts/profile_all_kernels.sh); factor 2 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 128-B "
                         "requests as 64 B; Infinity-Cache hits are counted).  bench.py reports a number only while kernel_source_sha equals "
                         "the hash of that kernel's sources in the tree (bench.KERNEL_SOURCES)")
placer = {f"group{g}": bench_numbers(out / f"g{g}_stats_stdout.log").get("_memory_groups") for g in GROUPS}
(out / "all_kernels_summary.json").write_text(json.dumps({"tag": tag, "git_sha": git_sha, "memory_group_placer": placer, "kernels": summary}, indent=1))
(out / "hbm_traffic.json").write_text(json.dumps(traffic_file, indent=1))
for w, e in summary.items():
    print(f"{w:52s} {e.get('average_ns', 0) / 1e3:9.1f} us  x{e.get('traffic_over_algorithmic')}  frac {e.get('frac_of_hbm_peak')}  {e.get('error', '')}")
