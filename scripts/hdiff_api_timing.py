#!/usr/bin/env python3
"""Launch-by-launch HIP-event times of the f32 / f64 horizontal diffusion through the user API, for several placements
of the three fields (GT4PY_AMD_ALLOC_SKEW_BYTES): is the spread a property of the placement or of the device state?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import gt4py_amd.storage as gt_storage
from gt4py_amd.cartesian import gtscript
from gt4py_amd.cartesian.backend import hip_templates


def run(dt, dom, skew, lit=64, n=120):
    os.environ["GT4PY_AMD_ALLOC_SKEW_BYTES"] = str(skew)
    obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt}, device_sync=False,
                           literal_float_precision=lit)
    shape = (dom[0] + 4, dom[1] + 4, dom[2])
    gen = torch.Generator(device="cuda").manual_seed(1)
    fields = {}
    for name in ("in_field", "coeff", "out_field"):
        f = gt_storage.empty(shape, dt, backend="hip:mi300", aligned_index=(2, 2, 0))
        f.tensor.copy_(torch.rand(shape, dtype=f.tensor.dtype, device="cuda", generator=gen))
        fields[name] = f
    frozen = obj.freeze(origin={k: (2, 2, 0) for k in fields}, domain=dom)
    for i in range(5):
        frozen(**fields)
    torch.cuda.synchronize()
    t = bench._time_launches(lambda i: frozen(**fields), n)
    # back-to-back, one event pair around all launches (what the micro-benchmark does)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        frozen(**fields)
    b.record()
    b.synchronize()
    lups = float(np.prod(dom))
    print(f"{np.dtype(dt).name} lit{lit} skew={skew:8d} addr%4MiB={[int(f.ptr % (4 << 20)) >> 10 for f in fields.values()]} KiB  "
          f"median {t['median']:.4f} min {t['min']:.4f} max {t['max']:.4f} mean {t['mean']:.4f} ms | back-to-back {a.elapsed_time(b) / n:.4f} ms "
          f"-> {lups / t['median'] / 1e6:.1f} / {lups / (a.elapsed_time(b) / n) / 1e6:.1f} GLUPS", flush=True)


for rep in range(2):
    for skew in (1 << 20, 0, 3 << 19, 1 << 19):
        run(np.float32, (1024, 1024, 80), skew)
    run(np.float32, (1024, 1024, 80), 1 << 20, lit=32)
    for skew in (1 << 20, 0):
        run(np.float64, (512, 1024, 80), skew)
