#!/usr/bin/env python3
"""Run on the GPU box: A/B of the float64 horizontal-diffusion strips (6 rows / 6 in flight, the default since round 5, against 8 / 8)
on ONE box in the product's call path: alternating child processes, bench.other_kernels on the configs[4] share (512 x 1024 x 80).

    python3 scripts/hdiff_f64_strip_ab.py [--rounds 4] >> profiles/r5_hdiff_f32_strip_ab.log"""
import argparse
import json
import os
import pathlib
import statistics
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
CHILD = ("import sys, json; sys.path.insert(0, %r); import torch; torch.cuda.set_device(0); import bench; "
         "out = bench.other_kernels(steps=300, only={'hdiff_limiter_f64_512x1024x80'}); "
         "print(json.dumps({k: v['ms'] for k, v in out.items()}))") % str(ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    seen = {6: [], 8: []}
    for r in range(args.rounds):
        for rows in (6, 8) if r % 2 == 0 else (8, 6):
            proc = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, GT4MI_HDIFF_F64_ROWS=str(rows)), capture_output=True, text=True, timeout=600)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(f"round {r} rows {rows}: FAILED\n{proc.stderr[-2000:]}")
                continue
            ms = json.loads(line[-1])["hdiff_limiter_f64_512x1024x80"]
            seen[rows].append(ms)
            print(f"round {r}  rows per strip {rows}:  hdiff_limiter_f64_512x1024x80 {ms:.4f} ms  {24.0 * 512 * 1024 * 80 / (ms * 1e-3) / 8e12:.4f} of 8 TB/s", flush=True)
    if seen[6] and seen[8]:
        a, b = statistics.median(seen[6]), statistics.median(seen[8])
        print(f"median  6 rows / 6 in flight {a:.4f} ms   8 rows / 8 in flight {b:.4f} ms   ({(b / a - 1) * 100:+.2f} % for 8 / 8)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
