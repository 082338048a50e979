#!/usr/bin/env python3
"""Run on the GPU box: rolling-prefetch distance of the generated column kernels' register levels (GT4MI_CODEGEN_TOP_CACHE_LOOKAHEAD) with
the nontemporal loads, ONE process, SAME fields.   python3 scripts/column_lookahead_ab.py"""
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_codegen  # noqa: E402
from gt4py_amd.storage import placement  # noqa: E402

CASES = (("lookahead", 0), ("lookahead", 2), ("lookahead", 6), ("lookahead", 8), ("prefetch", 4), ("prefetch", 16))


def main() -> int:
    torch.cuda.set_device(0)
    placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=5)
    gen = torch.Generator(device="cuda").manual_seed(3)
    dom = (1024, 1024, 160)
    shape = (dom[0] + 1, dom[1], dom[2] + 1)

    def field():
        f = gt_storage.empty(shape, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0))
        f.tensor.copy_(torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1)
        return f

    vf = {n: field() for n in ("utens_stage", "u_stage", "wcon", "u_pos", "utens")}
    frozen = {}
    for what, value in CASES:
        key = "top_cache_lookahead" if what == "lookahead" else "prefetch"
        saved = hip_codegen.TUNING[key]
        hip_codegen.TUNING[key] = value
        try:
            st = gtscript.stencil(backend="hip:mi300", definition=bench._vertical_advection_dycore, externals={"BET_M": 0.5, "BET_P": 0.5}, device_sync=False,
                                  rebuild=True, name=f"vadv_{what}_{value}")
        finally:
            hip_codegen.TUNING[key] = saved
        frozen[(what, value)] = st.freeze(origin={k: (0, 0, 0) for k in vf}, domain=dom)
    for rep in range(4):
        row = []
        for key, fz in frozen.items():
            for _ in range(3):
                fz(**vf, dtr_stage=0.15)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fz(**vf, dtr_stage=0.15)
            b.record()
            b.synchronize()
            ms = a.elapsed_time(b) / 20
            row.append(f"{key[0]} {key[1]}: {ms:.4f} {48.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f}")
        print("   ".join(row), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
