#!/usr/bin/env python3
"""Run on the GPU box: nontemporal LOADS in the column kernels, off / on, alternating child processes on one box (every process
allocates its fields afresh through the storage layer, memory-group placer on as in bench.py).

    python3 scripts/nt_loads_ab.py > profiles/r5_nt_loads_ab.log

Child: `--child` prints one JSON line of bench.other_kernels() for the column workloads under the environment it was given."""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
WORKLOADS = ("tridiagonal_f64_1024x1024x160", "generated_vertical_advection_f64_1024x1024x160")


def child() -> int:
    import torch

    import bench
    from gt4py_amd.storage import placement

    torch.cuda.set_device(0)
    placement.configure(max_candidates=24, spacer_bytes=8 << 30, park_extra=5)
    out = bench.other_kernels(20, only=set(WORKLOADS))
    # the generated tridiagonal solve (the code generator's column kernel on the same problem as the hand-written one)
    import numpy as np

    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    dom = (1024, 1024, 160)
    tri = gtscript.stencil(backend="hip:mi300", definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64}, device_sync=False,
                           use_kernel_library=False)
    gen = torch.Generator(device="cuda").manual_seed(7)
    ranges = {"inf": (-1, 1), "diag": (4, 5), "sup": (-1, 1), "rhs": (-10, 10), "out": (0, 0)}
    host = {n: (torch.rand(dom, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo) for n, (lo, hi) in ranges.items()}
    fields = {n: gt_storage.empty(dom, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0)) for n in ranges}
    frozen = tri.freeze(origin={k: (0, 0, 0) for k in fields}, domain=dom)
    times = []
    for _ in range(8):
        for n in ranges:
            fields[n].tensor.copy_(host[n])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        frozen(**fields)
        b.record()
        b.synchronize()
        times.append(a.elapsed_time(b))
    ms = sorted(times)[len(times) // 2]
    line = {k: {"ms": v["ms"], "frac": v["frac_of_hbm_peak"], "classes": v.get("memory_classes")} for k, v in out.items()}
    line["generated_tridiagonal_f64_1024x1024x160"] = {"ms": round(ms, 4), "frac": round(56.0 * np.prod(dom) / (ms * 1e-3) / 8e12, 4),
                                                        "classes": [placement.class_of(f) for f in fields.values()]}
    print("RESULT " + json.dumps(line))
    return 0


def main() -> int:
    if "--child" in sys.argv:
        return child()
    cases = [("plain loads (one wave per workgroup)", {"GT4MI_TRIDIAG_NT_LOADS": "0", "GT4MI_CODEGEN_COLUMN_NT_LOADS": "0"}),
             ("nontemporal loads, one wave per workgroup; generated: mode 3", {"GT4MI_TRIDIAG_WPB": "1", "GT4MI_CODEGEN_COLUMN_NT_LOADS": "3"}),
             ("nontemporal loads, two waves per workgroup; generated: mode 5 (defaults)", {})]
    for rep in range(2):
        for what, env in cases:
            p = subprocess.run([sys.executable, __file__, "--child"], env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
            got = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if not got:
                print(f"{what}: FAILED rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}", flush=True)
                continue
            r = json.loads(got[-1][7:])
            print(f"{what:58s} " + "   ".join(f"{k.split('_f64')[0]} {v['ms']:.4f} ms {v['frac']:.4f}" for k, v in r.items()), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
