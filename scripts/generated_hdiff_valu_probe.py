"""Is the generated (strip-kernel) horizontal diffusion bound by arithmetic?  The same float32 stencil with float64
literals (the reference's default: lap / flx / fly are float64) and with literal_float_precision=32 (all float32, half the
VALU work per point), generated vs kernel library, 1024 x 1024 x 80.

    python scripts/generated_hdiff_valu_probe.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_templates  # noqa: E402

dom = (1024, 1024, 80)
for prec in (64, 32):
    for use_lib in (True, False):
        obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float32},
                               device_sync=False, use_kernel_library=use_lib, literal_float_precision=prec)
        shape = (dom[0] + 4, dom[1] + 4, dom[2])
        f = {n: gt_storage.ones(shape, np.float32, backend="hip:mi300", aligned_index=(2, 2, 0)) for n in ("in_field", "out_field", "coeff")}
        for v in f.values():
            v.tensor.uniform_(-1.0, 1.0)
        frozen = obj.freeze(origin={n: (2, 2, 0) for n in f}, domain=dom)
        for _ in range(300):
            frozen(**f)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300):
            frozen(**f)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 300
        print(f"literal_float_precision={prec} {'library  ' if use_lib else 'generated'} {ms:.4f} ms  {np.prod(dom) / ms / 1e6:6.1f} GLUPS "
              f"{np.prod(dom) * 12 / ms / 1e9:5.2f} TB/s", flush=True)
