#!/usr/bin/env python3
"""Why does bench.py report the f32 horizontal diffusion ~10 % slower than a stand-alone timing?  Same stencil, same
fields, timed (a) in a fresh process, (b) after the Laplacian fields exist, (c) after the Laplacian ran, (d) after the
streaming copy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import gt4py_amd.storage as gt_storage
from gt4py_amd.cartesian import gtscript
from gt4py_amd.cartesian.backend import hip_templates

torch.cuda.set_device(0)
gen = torch.Generator(device="cuda").manual_seed(2024)


def hdiff_case(tag, ranges=((0.0, 10.0), (0.0, 0.05), (-1.0, 1.0))):
    dt, dom = np.float32, (1024, 1024, 80)
    obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt}, device_sync=False)
    shape = (dom[0] + 4, dom[1] + 4, dom[2])

    def field(lo, hi):
        f = gt_storage.empty(shape, dt, backend="hip:mi300", aligned_index=(2, 2, 0))
        f.tensor.copy_(torch.rand(shape, dtype=f.tensor.dtype, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    fields = {"in_field": field(*ranges[0]), "coeff": field(*ranges[1]), "out_field": field(*ranges[2])}
    frozen = obj.freeze(origin={k: (2, 2, 0) for k in fields}, domain=dom)
    for i in range(10):
        frozen(**fields)
    torch.cuda.synchronize()
    t = bench._time_launches(lambda i: frozen(**fields), 20)
    print(f"{tag:40s} mean {t['mean']:.4f} median {t['median']:.4f} min {t['min']:.4f} ms  addr%4MiB(KiB)="
          f"{[int(f.ptr % (4 << 20)) >> 10 for f in fields.values()]} ptrs={[hex(f.ptr) for f in fields.values()]}", flush=True)


hdiff_case("fresh process")
hdiff_case("fresh process, again")
unit = ((0.0, 1.0), (0.0, 1.0), (0.0, 1.0))
hdiff_case("all fields in [0, 1)", unit)
hdiff_case("all fields in [0, 1), again", unit)
hdiff_case("in [1, 9), coeff [0, 0.05)", ((1.0, 9.0), (0.0, 0.05), (0.0, 1.0)))
hdiff_case("in [0, 10), coeff [0, 1)", ((0.0, 10.0), (0.0, 1.0), (0.0, 1.0)))
hdiff_case("in [0, 1), coeff [0, 0.05)", ((0.0, 1.0), (0.0, 0.05), (0.0, 1.0)))
hdiff_case("bench ranges once more")
pairs = bench._device_fields((514, 514, 512), n_pairs=2, seed=1337)
hdiff_case("after allocating the Laplacian fields")
lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64}, device_sync=False)
frozen = lap.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=(512, 512, 512))
for i in range(100):
    frozen(inp=pairs[i % 2][0], out=pairs[i % 2][1])
torch.cuda.synchronize()
hdiff_case("after 100 Laplacian applies")
print("copy", round(bench.copy_ceiling_gbs(), 1))
hdiff_case("after the streaming copy")
hdiff_case("after the streaming copy, again")
