#!/usr/bin/env python3
"""Run on the GPU box: a random campaign of horizontal diffusion through the C ABI against the oracle -- widths and heights around the
partition points of hdiff_share_kernel (strips of 62 lanes x 16 bytes, workgroups of 4 waves x 4 rows), origins 0-3 items off a 16-byte
boundary, both dtypes and internal precisions, limiter on / off, field / scalar coefficient; the halo and everything outside the domain
must stay untouched (NaN canaries in the padding, -7 in the halo).

(Lives under tests/ because it calls the oracle: only tests, smoke() and bench.py's cpu_baseline leg may.)

    python3 tests/hdiff_campaign.py [--cases 600] [--seed 6] > profiles/r6_hdiff_share_campaign.log"""
import argparse
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))  # gpu_util
import gpu_util as G  # noqa: E402
from gt4py_amd import _lib  # noqa: E402
from oracle import ref_numpy as R  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=600)
    ap.add_argument("--seed", type=int, default=6)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    widths = [32, 33, 61, 62, 63, 123, 124, 125, 126, 127, 186, 247, 248, 249, 250, 251, 372, 496, 497, 600]
    for case in range(args.cases):
        dtype = np.float64 if rng.integers(2) else np.float32
        dI = int(rng.choice(widths)) if rng.integers(3) else int(rng.integers(32, 700))
        dJ = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 47, 48, 49, 65])) if rng.integers(3) else int(rng.integers(1, 90))
        dK = int(rng.integers(1, 4))
        align = (int(rng.integers(0, 4)), int(rng.integers(0, 3)), 0)
        limiter, cf_field = bool(rng.integers(4)), bool(rng.integers(3))
        lit32 = bool(dtype == np.float32 and rng.integers(3) == 0)
        shape = (dI + 4, dJ + 4, dK)
        u = rng.uniform(-10, 10, shape).astype(dtype)
        c = rng.uniform(0, 0.5, shape).astype(dtype)
        weight = c if cf_field else (np.float32(0.31) if lit32 else np.float64(0.31))
        want = np.full(shape, -7.0, dtype)
        R.hdiff(u, want, weight, origin_in=(2, 2, 0), origin_out=(2, 2, 0), origin_coeff=(2, 2, 0), domain=(dI, dJ, dK), limiter=limiter,
                literal_float_precision=32 if lit32 else 64)
        flags = (_lib.HDIFF_LIMITER if limiter else 0) | (_lib.HDIFF_INTERNAL_F32 if lit32 else 0) | (_lib.HDIFF_COEFF_F32 if (lit32 and not cf_field) else 0)
        d_u = G.DevArray(u, "ifirst", align)
        d_c = G.DevArray(c, "ifirst", align) if cf_field else float(weight)
        d_o = G.DevArray(np.full(shape, -7.0, dtype), "ifirst", align)
        G.hdiff(d_u, d_o, d_c, (2, 2, 0), (2, 2, 0), (2, 2, 0) if cf_field else None, (dI, dJ, dK), flags)
        got = d_o.get()
        same = np.array_equal(got, want)
        if not same:
            bad += 1
            print(f"case {case}: MISMATCH  {np.dtype(dtype).name} domain {(dI, dJ, dK)} aligned_index {align} limiter {limiter} coeff {'field' if cf_field else 'scalar'} "
                  f"literal32 {lit32}: {int((got != want).sum())} elements differ", flush=True)
    print(f"{args.cases} random cases (seed {args.seed}): {args.cases - bad} bit-identical to the oracle, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
