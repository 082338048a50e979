#!/usr/bin/env python3
"""Run on the GPU box: A/B of the float32 horizontal-diffusion strips on ONE box, in the product's call path.

    python3 scripts/hdiff_f32_strip_ab.py [--rounds 4] > profiles/r5_hdiff_f32_strip_ab.log

Alternates child processes with GT4MI_HDIFF_F32_ROWS=6 (6 rows per strip, all 6 in flight: the default) and =8 (8 rows, 4 in
flight: less traffic at the memory side) and prints, per process, what bench.other_kernels measures for BASELINE configs[2] with float64 and with float32
literals (HIP events, clocks settled).  Boxes differ by 1-3 %; the two strip shapes by less -- hence one box, alternating."""
import argparse
import json
import os
import pathlib
import statistics
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
CHILD = ("import sys, json; sys.path.insert(0, %r); import torch; torch.cuda.set_device(0); import bench; "
         "out = bench.other_kernels(steps=300, only={'hdiff_limiter_f32_1024x1024x80', 'hdiff_limiter_f32_literal32_1024x1024x80'}); "
         "print(json.dumps({k: v['ms'] for k, v in out.items()}))") % str(ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    seen = {6: {}, 8: {}}
    for r in range(args.rounds):
        for rows in (6, 8) if r % 2 == 0 else (8, 6):
            proc = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, GT4MI_HDIFF_F32_ROWS=str(rows)), capture_output=True, text=True,
                                  timeout=600)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
            if proc.returncode != 0 or not line:
                print(f"round {r} rows {rows}: FAILED {proc.stderr[-400:]}")
                continue
            ms = json.loads(line[-1])
            print(f"round {r}  rows per strip {rows}:  " + "  ".join(f"{k} {v:.4f} ms" for k, v in ms.items()), flush=True)
            for k, v in ms.items():
                seen[rows].setdefault(k, []).append(v)
    for k in seen[6]:
        a, b = statistics.median(seen[6][k]), statistics.median(seen[8].get(k, [float('nan')]))
        print(f"median  {k}:  6 rows / 6 in flight {a:.4f} ms   8 rows / 4 in flight {b:.4f} ms   ({100.0 * (b - a) / a:+.2f} %)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
