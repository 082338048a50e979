#!/usr/bin/env python3
"""Run on the GPU box: workgroup shape of the GENERATED column kernels (4 / 2 / 1 waves per workgroup, the LDS share scaled so that a
CU still holds four waves) with the nontemporal loads, in ONE process on the SAME fields.

    python3 scripts/column_block_ab.py >> profiles/r5_nt_loads_column_kernels.txt"""
import os
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_codegen, hip_templates  # noqa: E402
from gt4py_amd.storage import placement  # noqa: E402

SHAPES = ((4, 160), (2, 80), (1, 40))  # (waves per workgroup, KB of LDS per workgroup)


def variants(definition, **kw):
    saved = (hip_codegen.TUNING["block_column"], hip_codegen.TUNING["top_cache"])
    out = {}
    try:
        for waves, kb in SHAPES:
            hip_codegen.TUNING["block_column"] = (64, waves)
            hip_codegen.TUNING["top_cache"] = (-1, kb * 1024, 64)
            out[waves] = gtscript.stencil(backend="hip:mi300", definition=definition, device_sync=False, rebuild=True, name=f"{definition.__name__}_w{waves}", **kw)
    finally:
        hip_codegen.TUNING["block_column"], hip_codegen.TUNING["top_cache"] = saved
    return out


def main() -> int:
    torch.cuda.set_device(0)
    placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=5)
    gen = torch.Generator(device="cuda").manual_seed(3)
    dom = (1024, 1024, int(os.environ.get("COLUMN_K", "160")))

    def field(shape, lo=-1.0, hi=1.0):
        f = gt_storage.empty(shape, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0))
        f.tensor.copy_(torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    shape = (dom[0] + 1, dom[1], dom[2] + 1)
    vf = {n: field(shape) for n in ("utens_stage", "u_stage", "wcon", "u_pos", "utens")}
    vadv = variants(bench._vertical_advection_dycore, externals={"BET_M": 0.5, "BET_P": 0.5})
    frozen = {m: s.freeze(origin={k: (0, 0, 0) for k in vf}, domain=dom) for m, s in vadv.items()}
    print(f"generated vertical advection {dom}, waves per workgroup", flush=True)
    for rep in range(4):
        row = []
        for m in frozen:
            for _ in range(3):
                frozen[m](**vf, dtr_stage=3.0 / 20.0)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                frozen[m](**vf, dtr_stage=3.0 / 20.0)
            b.record()
            b.synchronize()
            ms = a.elapsed_time(b) / 20
            row.append(f"{m} waves: {ms:.4f} ms {48.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f}")
        print("   ".join(row), flush=True)
    del vf, frozen
    torch.cuda.empty_cache()
    ranges = {"inf": (-1, 1), "diag": (4, 5), "sup": (-1, 1), "rhs": (-10, 10), "out": (0, 0)}
    tf = {n: field(dom, *r) for n, r in ranges.items()}
    host = {n: f.tensor.clone() for n, f in tf.items()}
    tri = variants(hip_templates.tridiagonal_solver, dtypes={"T": np.float64}, use_kernel_library=False)
    frozen = {m: s.freeze(origin={k: (0, 0, 0) for k in tf}, domain=dom) for m, s in tri.items()}
    print(f"generated tridiagonal solve {dom}, waves per workgroup", flush=True)
    for rep in range(4):
        row = []
        for m in frozen:
            times = []
            for _ in range(6):
                for n in ranges:
                    tf[n].tensor.copy_(host[n])
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                frozen[m](**tf)
                b.record()
                b.synchronize()
                times.append(a.elapsed_time(b))
            ms = sorted(times)[len(times) // 2]
            row.append(f"{m} waves: {ms:.4f} ms {56.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f}")
        print("   ".join(row), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
