#!/usr/bin/env python3
"""Run on the GPU box, under rocprofv3 (scripts/profile_all_kernels.sh): every kernel of BASELINE.json at its own size through the
product's call path -- the headline Laplacian 512^3 and the entries of `bench.other_kernels()` -- a few launches each.

    python3 scripts/run_baseline_kernels.py [--only name,name,...] [--steps 10]

`--only`: keys of bench.KERNEL_SOURCES (default: all).  Prints one JSON object: what bench.py itself measured for each (HIP events),
next to which the profiler's per-kernel averages must agree.
"""
import argparse
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    only = set(n for n in args.only.split(",") if n) or set(bench.KERNEL_SOURCES)
    unknown = only - set(bench.KERNEL_SOURCES)
    if unknown:
        raise SystemExit(f"unknown workload(s) {sorted(unknown)}; known: {sorted(bench.KERNEL_SOURCES)}")
    import numpy as np
    import torch

    from gt4py_amd.cartesian import gtscript

    torch.cuda.set_device(0)
    from gt4py_amd.storage import placement

    # (as bench.py: the allocator's wide search for a second memory group)
    placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=8)
    out = {}
    if "lap5_f64_512" in only:  # the headline workload exactly as bench.py's N = 1 line runs it
        lap = gtscript.stencil(backend="hip:mi300", definition=bench._lap_definition(), dtypes={"T": np.float64}, device_sync=False)
        shape = (bench.GRID[0] + 2, bench.GRID[1] + 2, bench.GRID[2])
        pairs = bench._device_fields(shape, n_pairs=2, seed=1337, hint=lap.placement_hint())
        frozen = lap.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=bench.GRID)

        def step(i):
            inp, o = pairs[i % len(pairs)]
            frozen(inp=inp, out=o)

        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t = bench._time_launches(step, max(args.steps, 10))
        lups = float(np.prod(bench.GRID))
        out["lap5_f64_512"] = {"ms": round(t["mean"], 5), "glups": round(lups / t["mean"] / 1e6, 1),
                               "frac_of_hbm_peak": round(16.0 * lups / (t["mean"] * 1e-3) / 1e9 / bench.PEAK_GBS, 4)}
        del pairs
        torch.cuda.empty_cache()
    rest = only - {"lap5_f64_512"}
    if rest:
        for name, entry in bench.other_kernels(steps=args.steps, only=rest).items():
            out[name] = {k: entry[k] for k in ("ms", "glups", "frac_of_hbm_peak") if k in entry}
    out["_memory_groups"] = {k: v for k, v in (placement.report() or {}).items() if k != "fields"}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
