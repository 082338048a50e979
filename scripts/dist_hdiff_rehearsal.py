"""1-GPU rehearsal of BASELINE config[4]: horizontal diffusion 2048x2048x80 fp64 over a 4x2 process grid.

One rank's share is 512x1024x80 with ghost zones 2 deep on all four sides; here the four neighbours are
the rank itself (periodic self-loop through RCCL), so the choreography -- fork, interior kernel, pack ->
send/recv -> unpack on the side stream, join, four boundary strips -- runs exactly as it would on 8 GPUs,
only the wire is an on-device copy.  Prints the per-apply time next to the undecomposed kernel's.

    python scripts/dist_hdiff_rehearsal.py [--iters 50]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import gt4py_amd.storage as gt_storage  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402
from gt4py_amd.cartesian.backend import hip_templates  # noqa: E402
from gt4py_amd.distributed import Decomposition, NativeComm, NativeHaloExchanger, overlapped_apply  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64},
                          device_sync=False)
    local = (512, 1024, 80)
    dec = Decomposition(local, (1, 1), 0, 2, periodic=(True, True))
    comm = NativeComm(rank=0, world_size=1)
    gen = torch.Generator(device="cuda").manual_seed(2024)

    def field(lo, hi):
        f = gt_storage.empty(dec.local_shape, np.float64, backend="hip:mi300", aligned_index=dec.origin)
        f.tensor.copy_(torch.rand(dec.local_shape, dtype=torch.float64, device="cuda", generator=gen) * (hi - lo) + lo)
        return f

    fields = {"in_field": field(0.0, 10.0), "coeff": field(0.0, 0.05), "out_field": field(0.0, 0.0)}
    origin = {n: dec.origin for n in fields}
    ex = NativeHaloExchanger(dec, np.float64, comm)
    frozen = hd.freeze(origin=origin, domain=local)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(args.iters):
            fn()
        stop.record()
        torch.cuda.synchronize()
        return start.elapsed_time(stop) / args.iters

    t_kernel = timed(lambda: frozen(**fields))
    t_dist = timed(lambda: overlapped_apply(hd, dec, origin, fields, {"in_field": ex}))
    t_seq = timed(lambda: (ex.exchange(fields["in_field"]), frozen(**fields)))
    lups = float(np.prod(local))
    print(f"hdiff fp64 512x1024x80 (one rank of 4x2 over 2048x2048x80), ghost depth 2, {ex.bytes_per_exchange / 1e6:.2f} MB exchanged per apply")
    print(f"  kernel alone                      {t_kernel:7.4f} ms  {lups / t_kernel / 1e6:6.1f} GLUPS per GPU")
    print(f"  exchange, then kernel (no overlap) {t_seq:7.4f} ms  {lups / t_seq / 1e6:6.1f}")
    print(f"  overlapped apply                  {t_dist:7.4f} ms  {lups / t_dist / 1e6:6.1f}  -> weak-scaling efficiency "
          f"{t_kernel / t_dist:.2f} if the links keep up")


if __name__ == "__main__":
    main()
