#!/usr/bin/env python3
"""Exchange-per-apply Laplacian step on the 1-GPU self-loop, shares of 8 ranks (1x8, 2x4, 4x2): us per apply for every message table x
schedule (join / chain / swap) x interior throttle -- the sweep behind DESIGN.md section 6's "swap" paragraph."""
import sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch, bench
from gt4py_amd.distributed.halo import Decomposition
from gt4py_amd.distributed.native import NativeComm, NativeHaloExchanger
def timed(fn, n):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
comm = NativeComm(rank=0, world_size=1)
for g in ((1, 8), (2, 4), (4, 2)):
    total = (512 // g[0], 512 // g[1], 512); periodic = (g[0] > 1, g[1] > 1)
    dec = Decomposition(total, (1, 1), 0, halo=1, periodic=periodic)
    for rep in range(2):
      for single in (True, False):
        for sched in ("join", "chain", "swap"):
            for wg in (0, 3, 2):
                pairs = bench._device_fields(dec.local_shape, n_pairs=2, seed=1, origin=dec.origin)
                exs = [NativeHaloExchanger(dec, np.float64, comm, single_phase=single).tune(sched, wg) for _ in pairs]
                bound = [ex.make_dist_lap5(i, o, dec.origin, dec.origin) for ex, (i, o) in zip(exs, pairs)]
                st = {"i": 0}
                def call():
                    bound[st["i"] % 2](); st["i"] += 1
                us = timed(call, 200)
                print(f"{g} {'single' if single else 'two'} {sched:5s} wg{wg}: {us:6.1f} us per apply", flush=True)
                for ex in exs: ex.close()
                del bound, exs, pairs
