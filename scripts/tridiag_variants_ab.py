#!/usr/bin/env python3
"""Run on the GPU box: the variants of the deep tridiagonal kernel in ONE process on the SAME fields (placed by the storage layer, memory
-group placer on): plain loads / nontemporal loads with one wave per workgroup / with two (GT4MI_TRIDIAG_AB=1 makes the library read
its switches at every call).

    GT4MI_TRIDIAG_AB=1 python3 scripts/tridiag_variants_ab.py >> profiles/r5_nt_loads_column_kernels.txt"""
import os
import pathlib
import sys

os.environ["GT4MI_TRIDIAG_AB"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_templates  # noqa: E402
from gt4py_amd.storage import placement  # noqa: E402

VARIANTS = (("plain loads, 1 wave / workgroup", {"GT4MI_TRIDIAG_NT_LOADS": "0", "GT4MI_TRIDIAG_WPB": "1"}),
            ("nontemporal loads, 1 wave / workgroup", {"GT4MI_TRIDIAG_NT_LOADS": "1", "GT4MI_TRIDIAG_WPB": "1"}),
            ("nontemporal loads, 2 waves / workgroup", {"GT4MI_TRIDIAG_NT_LOADS": "1", "GT4MI_TRIDIAG_WPB": "2"}))


def main() -> int:
    torch.cuda.set_device(0)
    placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=5)
    dom = (1024, 1024, 160)
    gen = torch.Generator(device="cuda").manual_seed(7)
    ranges = {"inf": (-1, 1), "diag": (4, 5), "sup": (-1, 1), "rhs": (-10, 10), "out": (0, 0)}
    tri = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64}, device_sync=False)
    for aset in range(2):  # two allocation sets
        fields = {n: gt_storage.empty(dom, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0)) for n in ranges}
        host = {n: (torch.rand(dom, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo) for n, (lo, hi) in ranges.items()}
        frozen = tri.freeze(origin={k: (0, 0, 0) for k in fields}, domain=dom)
        print(f"allocation set {aset}: classes {[placement.class_of(f) for f in fields.values()]}", flush=True)
        outs = {}
        for rep in range(4):
            row = []
            for what, env in VARIANTS:
                os.environ.update(env)
                times = []
                for _ in range(6):
                    for n in ranges:
                        fields[n].tensor.copy_(host[n])
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    frozen(**fields)
                    b.record()
                    b.synchronize()
                    times.append(a.elapsed_time(b))
                ms = sorted(times)[len(times) // 2]
                outs[what] = fields["out"].tensor.clone()
                row.append(f"{what}: {ms:.4f} ms {56.0 * np.prod(dom) / (ms * 1e-3) / 8e12:.4f}")
            print("   ".join(row), flush=True)
        first = next(iter(outs.values()))
        print("   bit-identical results:", all(torch.equal(first, o) for o in outs.values()), flush=True)
        del fields, frozen, host
    return 0


if __name__ == "__main__":
    sys.exit(main())
