"""Time the generic executor's generated kernels next to the hand-written library (MI355X).

    python scripts/bench_generic.py [--iters 20]

Prints one line per case: ms per call (HIP events around back-to-back frozen calls), GLUPS, and the
bandwidth the algorithmic bytes of SURVEY.md section 8d correspond to.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

import gt4py_amd.storage as gt_storage  # noqa: E402
import stencil_zoo as zoo  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402

CASES = [
    # name, domain, algorithmic bytes per lattice update, use_kernel_library values to compare
    ("laplacian", (512, 512, 512), 16, (True, False)),
    ("horizontal_diffusion", (512, 1024, 80), 24, (True, False)),
    ("horizontal_diffusion_f32", (1024, 1024, 80), 12, (True, False)),
    ("horizontal_diffusion_if", (512, 1024, 80), 24, (False,)),  # the limiter as if / else blocks
    ("hyperdiffusion_6th", (512, 1024, 80), 16, (False,)),  # three nested Laplacians: 1 array read, 1 written
    ("tridiagonal_solver", (1024, 1024, 160), 56, (True, False)),
    ("tridiagonal_solver", (1024, 1024, 80), 56, (True, False)),
    ("tridiagonal_solver", (1024, 1024, 60), 56, (True, False)),
    ("vertical_advection_dycore", (1024, 1024, 160), 48, (False,)),  # 5 reads + 1 write; temporaries extra
    ("vertical_advection_dycore", (1024, 1024, 80), 48, (False,)),  # the usual number of levels of a regional model
    ("vertical_advection_dycore", (1024, 1024, 60), 48, (False,)),
    ("column_sum_then_gradient", (1024, 1024, 80), 16, (False,)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    for name, domain, bytes_per_lup, modes in CASES:
        if args.only and args.only not in name:
            continue
        defn, externals, scalars, _ = zoo.ZOO[name]
        for use_lib in modes:
            obj = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, device_sync=False,
                                   use_kernel_library=use_lib)
            arrays, origins = zoo.make_inputs(obj, (4, 4, max(4, obj.domain_info.min_sequential_axis_size)))
            dev = {}
            for k, v in arrays.items():
                info = obj.field_info[k]
                shape = tuple(d + max(0, b[0]) + max(0, b[1]) for d, b in zip(domain, info.boundary))
                dev[k] = gt_storage.ones(shape, v.dtype, backend="hip:mi300", aligned_index=origins[k])
                dev[k].tensor.uniform_(-1.0, 1.0) if v.dtype.kind == "f" else None
                if k == "diag":
                    dev[k].tensor.add_(4.5)
            if os.environ.get("GT4MI_PRINT_PTRS"):
                print("   field addresses mod 16 MiB (MiB):", {k: round((v.ptr % (1 << 24)) / 2**20, 3) for k, v in dev.items()},
                      " full:", {k: hex(v.ptr) for k, v in dev.items()})
            frozen = obj.freeze(origin=origins, domain=domain)
            for _ in range(3):
                frozen(**dev, **scalars)
            torch.cuda.synchronize()
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(args.iters):
                frozen(**dev, **scalars)
            stop.record()
            torch.cuda.synchronize()
            ms = start.elapsed_time(stop) / args.iters
            lups = float(np.prod(domain))
            kind = "library" if use_lib else "generated"
            print(f"{name:28s} {str(domain):18s} {kind:9s} {ms:8.3f} ms  {lups / ms / 1e6:7.1f} GLUPS  "
                  f"{lups * bytes_per_lup / ms / 1e9:6.2f} TB/s (algorithmic)", flush=True)
            del dev, frozen
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
