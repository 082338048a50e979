#!/usr/bin/env python3
"""The direct halo transport (peer stores from the pack kernel, csrc/direct.hip.h) on the 1-GPU self-loop: every form of the fused
steps checked on exactly known fields (distributed.FormCheck), and us per apply next to the RCCL transport -- Laplacian shares of 8
ranks (1x8, 2x4, 4x2) and BASELINE configs[4]'s share of horizontal diffusion."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_templates  # noqa: E402
from gt4py_amd.distributed import Decomposition, FormCheck, NativeComm, NativeHaloExchanger  # noqa: E402


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    comm = NativeComm(rank=0, world_size=1)
    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64}, device_sync=False)
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64}, device_sync=False)
    cases = [("lap", g) for g in ((1, 8), (2, 4), (4, 2))] + [("hdiff", (1, 1))]
    for what, g in cases:
        if what == "lap":
            total, periodic, halo = (512 // g[0], 512 // g[1], 512), (g[0] > 1, g[1] > 1), 1
        else:
            total, periodic, halo = (512, 1024, 80), (True, True), 2
        dec = Decomposition(total, (1, 1), 0, halo=halo, periodic=periodic)
        new = lambda: gt_storage.zeros(dec.local_shape, np.float64, backend="hip:mi300", aligned_index=dec.origin)  # noqa: E731
        if what == "lap":
            fr = lap.freeze(origin={"inp": dec.origin, "out": dec.origin}, domain=dec.local_domain)
            local = lambda a, b: fr(inp=a, out=b)  # noqa: E731
        else:
            coeff = new()
            coeff.tensor.fill_(0.025)
            fr = hd.freeze(origin={n: dec.origin for n in ("in_field", "out_field", "coeff")}, domain=dec.local_domain)
            local = lambda a, b: fr(in_field=a, out_field=b, coeff=coeff)  # noqa: E731
        chk = FormCheck(dec, new, local)
        kernel = timed(lambda: local(chk.probe, chk.out), 100)
        print(f"== {what} share {g}: local domain {dec.local_domain}, kernel alone {kernel:.1f} us", flush=True)
        schedules = ("join", "chain", "swap", "swap-packed", "inline")
        for transport in ("rccl", "direct"):
            for single in (True, False):
                best = {}
                for schedule in schedules:
                    for wg in (0, 3, 2):
                        pairs = [(new(), new()) for _ in range(2)]
                        for a, _ in pairs:
                            a.tensor.uniform_(-1, 1)
                        exs = [NativeHaloExchanger(dec, np.float64, comm, single_phase=single).tune(schedule, wg) for _ in range(3)]
                        if transport == "direct":
                            for ex in exs:
                                ex.use_direct_transport()

                        def make(ex, a, b):
                            if what == "lap":
                                return ex.make_dist_lap5(a, b, dec.origin, dec.origin)
                            return ex.make_dist_hdiff(a, b, coeff, dec.origin, type(hd)._gt_binding_.flags)

                        bound = [make(ex, a, b) for ex, (a, b) in zip(exs, pairs)]
                        probe = make(exs[2], chk.probe, chk.out)
                        chk.reset()
                        probe()
                        exs[2].end()
                        ok, found = chk.verdict()
                        st = {"i": 0}

                        def call():
                            bound[st["i"] % 2]()
                            st["i"] += 1

                        us = timed(call, 200)
                        status = exs[0].direct_status() if transport == "direct" else {}
                        if not ok or status.get("timed_out"):
                            print(f"   {transport} {'single' if single else 'two'} {schedule} wg{wg}: WRONG ({found}) {status}", flush=True)
                        best[schedule] = min(best.get(schedule, 1e9), us)
                        for ex in exs:
                            ex.close()
                        del bound, probe, exs, pairs
                print(f"   {transport:6s} {'single-phase' if single else 'two-phase   '}: " +
                      "  ".join(f"{s} {best[s]:6.1f}" for s in schedules) + "   us per apply (best throttle)", flush=True)
        del chk


if __name__ == "__main__":
    main()
