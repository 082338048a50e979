#!/usr/bin/env python3
"""Kernel time of the Laplacian on the per-rank shares of a 512^3 grid for every process grid of 2, 4, 8 ranks (1 GPU):
is a narrow local I extent (4x2: 128 columns) as efficient as a wide one (1x8: 512)?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from gt4py_amd.cartesian import gtscript
from gt4py_amd.cartesian.backend import hip_templates
from gt4py_amd.distributed import process_grid_candidates

lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64}, device_sync=False)
for n in (1, 2, 4, 8):
    for pi, pj in process_grid_candidates(n, (512, 512, 512)):
        dom = (512 // pi, 512 // pj, 512)
        for halo in (1, 2):
            shape = (dom[0] + 2 * halo, dom[1] + 2 * halo, dom[2])
            pairs = bench._device_fields(shape, n_pairs=4, seed=1, origin=(halo, halo, 0))
            frozen = lap.freeze(origin={"inp": (halo, halo, 0), "out": (halo, halo, 0)}, domain=dom)
            for i in range(20):
                frozen(inp=pairs[i % 4][0], out=pairs[i % 4][1])
            torch.cuda.synchronize()
            t = bench._time_launches(lambda i: frozen(inp=pairs[i % 4][0], out=pairs[i % 4][1]), 200)
            ideal = 16.0 * np.prod(dom) / 8e12 * 1e6
            print(f"{n} ranks, grid {pi}x{pj}, local {dom}, ghost depth {halo}: {t['mean'] * 1e3:7.1f} us per apply "
                  f"({16.0 * np.prod(dom) / (t['mean'] * 1e-3) / 1e9:6.0f} GB/s, {ideal / (t['mean'] * 1e3):.2f} of the HBM roofline)", flush=True)
            del pairs
            torch.cuda.empty_cache()
