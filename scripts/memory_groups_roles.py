#!/usr/bin/env python3
"""Run on the GPU box: which ASSIGNMENT of a stencil's fields to the two memory classes is fastest?  Fields allocated through
gt4py_amd.storage under `placement.want(cls)` (the placer's wide search), every assignment timed through the frozen call path.

    python3 scripts/memory_groups_roles.py > profiles/r5_memory_groups_roles.log"""
import itertools
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_templates  # noqa: E402
from gt4py_amd.storage import placement  # noqa: E402


def time_ms(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def main() -> int:
    torch.cuda.set_device(0)
    placer = placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=5)
    gen = torch.Generator(device="cuda").manual_seed(3)

    def field(shape, dtype, origin, cls, lo=-1.0, hi=1.0):
        with placement.want(cls):
            f = gt_storage.empty(shape, dtype, backend="hip:mi300", aligned_index=origin)
        f.tensor.copy_(torch.rand(shape, dtype=f.tensor.dtype, device="cuda", generator=gen) * (hi - lo) + lo)
        assert placement.class_of(f) in (cls, None) or placer.stats["wanted_class_not_found"], (placement.class_of(f), cls)
        return f

    # ---- horizontal diffusion fp64 share and fp32 configs[2] -------------------------------------------------------------------------
    for dt, dom in ((np.float64, (512, 1024, 80)), (np.float32, (1024, 1024, 80))):
        hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt}, device_sync=False)
        shape = (dom[0] + 4, dom[1] + 4, dom[2])
        for ci, cc, co in ((0, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 1, 0), (0, 0, 0), (0, 0, 1), (0, 1, 0)):
            fields = {"in_field": field(shape, dt, (2, 2, 0), ci, 4.0, 6.0), "coeff": field(shape, dt, (2, 2, 0), cc, 0.025, 0.025), "out_field": field(shape, dt, (2, 2, 0), co)}
            got = tuple(placement.class_of(f) for f in fields.values())
            frozen = hd.freeze(origin={k: (2, 2, 0) for k in fields}, domain=dom)
            ms = time_ms(lambda: frozen(**fields), 300)
            print(f"hdiff {np.dtype(dt).name} {dom}  in / coeff / out in classes {got}: {ms:.4f} ms  {3.0 * np.dtype(dt).itemsize * np.prod(dom) / (ms * 1e-3) / 8e12:.4f} of 8 TB/s", flush=True)
            del fields, frozen
    # ---- tridiagonal solve: inf diag (read) sup rhs (read + written) out (written) --------------------------------------------------------
    tri = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64}, device_sync=False)
    dom = (1024, 1024, 160)
    # (round 6) every assignment with inf in class 0, twice, then the mirror images of the best few
    every = [(0,) + c for c in itertools.product((0, 1), repeat=4)]
    for classes in every + every + [(1, 0, 1, 0, 1), (1, 0, 0, 1, 1), (1, 1, 0, 1, 0), (1, 0, 1, 0, 1), (1, 0, 0, 1, 1), (1, 1, 0, 1, 0)]:
        names = ("inf", "diag", "sup", "rhs", "out")
        ranges = {"inf": (-1, 1), "diag": (4, 5), "sup": (-1, 1), "rhs": (-10, 10), "out": (-1, 1)}
        fields = {n: field(dom, np.float64, (0, 0, 0), c, *ranges[n]) for n, c in zip(names, classes)}
        got = tuple(placement.class_of(f) for f in fields.values())
        frozen = tri.freeze(origin={k: (0, 0, 0) for k in fields}, domain=dom)
        ms = time_ms(lambda: frozen(**fields), 30)
        print(f"tridiagonal 1024x1024x160  inf / diag / sup / rhs / out in classes {got}: {ms:.4f} ms  {56.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f} of 8 TB/s", flush=True)
        del fields, frozen
    print({k: v for k, v in placement.report().items() if k != "fields"})
    return 0


if __name__ == "__main__":
    sys.exit(main())
