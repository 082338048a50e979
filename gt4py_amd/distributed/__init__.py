"""Multi-GPU execution of a stencil: IJ decomposition + halo exchange overlapped with compute.

NEW relative to the reference (gt4py.cartesian is single-device).
"""

from __future__ import annotations

from typing import Any, Dict, Mapping, Sequence

from .halo import Decomposition, HaloExchanger, HipPacker, choose_process_grid, halo_boxes, scatter_global
from .native import NativeComm, NativeHaloExchanger

__all__ = ["Decomposition", "HaloExchanger", "HipPacker", "NativeComm", "NativeHaloExchanger", "choose_process_grid",
           "halo_boxes", "overlapped_apply", "scatter_global"]


def _shifted(origin: Mapping[str, Sequence[int]], shift: Sequence[int]) -> Dict[str, tuple]:
    return {name: tuple(int(o) + int(s) for o, s in zip(org, shift)) for name, org in origin.items()}


def overlapped_apply(stencil, decomp: Decomposition, origin: Mapping[str, Sequence[int]], arguments: Dict[str, Any],
                     exchange: Mapping[str, HaloExchanger]) -> None:
    """One distributed apply of ``stencil`` on this rank.

    ``exchange`` maps the names of the read fields whose ghost cells are refreshed this step to
    their exchangers.  Timeline (two HIP streams):

        side stream : pack -> RCCL send/recv (I faces, then J faces) -> unpack
        main stream : interior kernel  ................  wait  -> boundary-strip kernels

    The stencil must have been built with ``device_sync=False`` so that launches stay asynchronous.
    ``arguments`` holds the device arrays / scalars by parameter name; ``origin`` the per-field origin
    of the LOCAL compute domain.
    """
    events = []
    native = [(name, ex) for name, ex in exchange.items() if isinstance(ex, NativeHaloExchanger)]
    for name, ex in exchange.items():
        if not isinstance(ex, NativeHaloExchanger):
            events.append((ex, ex.start(arguments[name].tensor)))
    # The exchange is enqueued BEFORE the interior kernel: its pack and send/recv kernels get onto the device
    # while it is idle.  Once the interior kernel saturates HBM, the few workgroups of a send/recv kernel see
    # loaded-memory latency and crawl (measured: 13 us alone, 170 us next to a 185 us hdiff interior), so
    # whatever finishes before that is a gain (profiles/r1_dist_hdiff_rehearsal.log).
    for name, ex in native:  # pack + RCCL + unpack on the side stream inside one C call
        ex.fork()
        ex.begin(arguments[name])
        events.append((ex, None))
    (shift, sub), strips = decomp.interior_and_strips()
    if all(d > 0 for d in sub):
        stencil.run(_domain_=tuple(sub), _origin_=_shifted(origin, shift), exec_info=None, **arguments)
    for ex, done in events:
        if done is None:
            ex.end()
        else:
            ex.finish(done)
    for shift, sub in strips:
        stencil.run(_domain_=tuple(sub), _origin_=_shifted(origin, shift), exec_info=None, **arguments)
