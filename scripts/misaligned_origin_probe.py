#!/usr/bin/env python3
"""What a compute-domain origin that is not on the storage's aligned column costs: the same stencils on storages allocated with
aligned_index = origin (the preset's intent) and with the default aligned_index = (0, 0, 0), and on odd domain widths."""
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
    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64}, device_sync=False)
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64}, device_sync=False)
    for dom in ((512, 512, 256), (511, 512, 256)):
        for aligned in ((1, 1, 0), (0, 0, 0)):
            shape = (dom[0] + 2, dom[1] + 2, dom[2])
            inp = gt_storage.ones(shape, backend="hip:mi300", aligned_index=aligned)
            out = gt_storage.zeros(shape, backend="hip:mi300", aligned_index=aligned)
            inp.tensor.uniform_(-1, 1)
            fr = lap.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=dom)
            ms = time_it(lambda: fr(inp=inp, out=out))
            print(f"lap5 f64 {dom} origin (1,1,0) aligned_index {aligned}: {ms:.4f} ms  {np.prod(dom) / ms / 1e6:7.1f} GLUPS")
    for dom in ((512, 1024, 80), (511, 1024, 80)):
        for aligned in ((2, 2, 0), (0, 0, 0), (1, 1, 0)):
            shape = (dom[0] + 4, dom[1] + 4, dom[2])
            f = {n: gt_storage.ones(shape, backend="hip:mi300", aligned_index=aligned) for n in ("in_field", "out_field", "coeff")}
            f["in_field"].tensor.uniform_(-1, 1)
            fr = hd.freeze(origin={n: (2, 2, 0) for n in f}, domain=dom)
            ms = time_it(lambda: fr(**f))
            print(f"hdiff f64 {dom} origin (2,2,0) aligned_index {aligned}: {ms:.4f} ms  {np.prod(dom) / ms / 1e6:7.1f} GLUPS")


if __name__ == "__main__":
    main()
