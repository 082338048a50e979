#!/usr/bin/env python3
"""Generated (hipRTC) kernels on storages whose compute origin is off the aligned column: horizontal diffusion (a `_vecs`
kernel: temporaries shared between lanes) and the Laplacian (a `_vec` kernel), float64 and float32."""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import gt4py_amd.storage as gt_storage
from gt4py_amd.cartesian import gtscript
from gt4py_amd.cartesian.backend import hip_templates


def time_it(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def main():
    for dt in (np.float64, np.float32):
        hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt},
                              device_sync=False, use_kernel_library=False)
        for dom in ((512, 1024, 80), (511, 1024, 80)):
            for aligned in ((2, 2, 0), (0, 0, 0), (1, 1, 0)):
                shape = (dom[0] + 4, dom[1] + 4, dom[2])
                f = {n: gt_storage.ones(shape, dtype=dt, backend="hip:mi300", aligned_index=aligned) for n in ("in_field", "out_field", "coeff")}
                f["in_field"].tensor.uniform_(-1, 1)
                fr = hd.freeze(origin={n: (2, 2, 0) for n in f}, domain=dom)
                ms = time_it(lambda: fr(**f))
                print(f"generated hdiff {np.dtype(dt).name} {dom} origin (2,2,0) aligned_index {aligned}: {ms:.4f} ms  "
                      f"{np.prod(dom) / ms / 1e6:7.1f} GLUPS", flush=True)


if __name__ == "__main__":
    main()
