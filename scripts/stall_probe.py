#!/usr/bin/env python3
"""Launch the kernels whose limiter is in question a few times each (for rocprofv3 --pmc passes; scripts/profile_stall_counters.sh):
horizontal diffusion fp64 512 x 1024 x 80 and fp32 1024 x 1024 x 80 (J-march kernel), with the fp64 Laplacian 512^3 as the
control that DOES reach the streaming-copy rate.  Through the storage layer and the stencil objects, like bench.py."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch

    import bench
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    gen = torch.Generator(device="cuda").manual_seed(1)

    def field(shape, dt, origin, lo=-1.0, hi=1.0):
        f = gt_storage.empty(shape, dt, backend="hip:mi300", aligned_index=origin)
        f.tensor.copy_(torch.rand(shape, dtype=f.tensor.dtype, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    for dt, dom in ((np.float64, (512, 1024, 80)), (np.float32, (1024, 1024, 80))):
        obj = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt}, device_sync=False)
        shape = (dom[0] + 4, dom[1] + 4, dom[2])
        fields = {"in_field": bench.hdiff_input(shape, dt, gen), "coeff": field(shape, dt, (2, 2, 0), 0.025, 0.025),
                  "out_field": field(shape, dt, (2, 2, 0))}
        frozen = obj.freeze(origin={k: (2, 2, 0) for k in fields}, domain=dom)
        for _ in range(n):
            frozen(**fields)
        torch.cuda.synchronize()
        del fields
    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64}, device_sync=False)
    dom = (512, 512, 512)
    shape = (dom[0] + 2, dom[1] + 2, dom[2])
    pairs = [{"inp": field(shape, np.float64, (1, 1, 0)), "out": field(shape, np.float64, (1, 1, 0))} for _ in range(2)]
    frozen = lap.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=dom)
    for i in range(n):
        frozen(**pairs[i % 2])
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
