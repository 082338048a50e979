#!/usr/bin/env python3
"""Run on the GPU box: A/B of horizontal diffusion's two fast paths on ONE box in the product's call path -- hdiff_share_kernel (waves
exchange their halo rows through LDS, the default since round 6) against the register-only J-march of rounds 1-5 (GT4MI_HDIFF_SHARE=0):
alternating child processes, bench.other_kernels (FrozenStencil, fields placed by role) on BASELINE configs[2] and the share of configs[4].

    python3 scripts/hdiff_share_ab.py [--rounds 4] > profiles/r6_hdiff_share_ab_product_path.log"""
import argparse
import json
import os
import pathlib
import statistics
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
NAMES = {"hdiff_limiter_f32_1024x1024x80": 12.0 * 1024 * 1024 * 80, "hdiff_limiter_f64_512x1024x80": 24.0 * 512 * 1024 * 80}
CHILD = ("import sys, json; sys.path.insert(0, %r); import torch; torch.cuda.set_device(0); import bench; "
         "from gt4py_amd.storage import placement; placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=8); "
         "out = bench.other_kernels(steps=300, only=%r); "
         "print(json.dumps({k: [v['ms'], v['memory_classes']] for k, v in out.items()}))") % (str(ROOT), set(NAMES))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--xcd", action="store_true", help="A/B the share kernel's XCD run length instead: 2 (library) against 4")
    args = ap.parse_args()
    if args.xcd:
        return xcd_ab(args.rounds)
    seen = {(n, s): [] for n in NAMES for s in (0, 1)}
    for r in range(args.rounds):
        for share in (1, 0) if r % 2 == 0 else (0, 1):
            proc = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, GT4MI_HDIFF_SHARE=str(share)), capture_output=True, text=True, timeout=600)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(f"round {r} share {share}: FAILED\n{proc.stderr[-2000:]}")
                continue
            for name, (ms, classes) in json.loads(line[-1]).items():
                seen[(name, share)].append(ms)
                print(f"round {r}  {'hdiff_share_kernel ' if share else 'hdiff_jmarch_kernel'}  {name}: {ms:.4f} ms  {NAMES[name] / (ms * 1e-3) / 8e12:.4f} of 8 TB/s  classes {classes}",
                      flush=True)
    for name in NAMES:
        a, b = seen[(name, 1)], seen[(name, 0)]
        if a and b:
            ma, mb = statistics.median(a), statistics.median(b)
            print(f"median {name}: share {ma:.4f} ms ({NAMES[name] / (ma * 1e-3) / 8e12:.4f})   J-march {mb:.4f} ms ({NAMES[name] / (mb * 1e-3) / 8e12:.4f})   "
                  f"({(ma / mb - 1) * 100:+.2f} % time for the share kernel)")
    return 0


def xcd_ab(rounds: int) -> int:
    seen = {(n, x): [] for n in NAMES for x in (2, 4)}
    for r in range(rounds):
        for runs in (2, 4) if r % 2 == 0 else (4, 2):
            proc = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, GT4MI_HDIFF_SHARE_XCD=str(runs)), capture_output=True, text=True, timeout=600)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(f"round {r} xcd runs {runs}: FAILED\n{proc.stderr[-2000:]}")
                continue
            for name, (ms, classes) in json.loads(line[-1]).items():
                seen[(name, runs)].append(ms)
                print(f"round {r}  XCD runs of {runs}  {name}: {ms:.4f} ms  {NAMES[name] / (ms * 1e-3) / 8e12:.4f} of 8 TB/s  classes {classes}", flush=True)
    for name in NAMES:
        a, b = seen[(name, 2)], seen[(name, 4)]
        if a and b:
            ma, mb = statistics.median(a), statistics.median(b)
            print(f"median {name}: runs of 2 {ma:.4f} ms   runs of 4 {mb:.4f} ms   ({(mb / ma - 1) * 100:+.2f} % time for runs of 4)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
