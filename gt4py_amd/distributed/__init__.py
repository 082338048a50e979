"""Multi-GPU execution of a stencil: IJ decomposition + halo exchange overlapped with compute.

NEW relative to the reference (gt4py.cartesian is single-device).
"""

from __future__ import annotations

from typing import Any, Dict, Mapping, Sequence

from .halo import Decomposition, HaloExchanger, HipPacker, choose_process_grid, scatter_global

__all__ = ["Decomposition", "HaloExchanger", "HipPacker", "choose_process_grid", "overlapped_apply", "scatter_global"]


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
    events = [(ex, ex.start(arguments[name].tensor)) for name, ex in exchange.items()]
    (shift, sub), strips = decomp.interior_and_strips()
    if all(d > 0 for d in sub):
        stencil.run(_domain_=tuple(sub), _origin_=_shifted(origin, shift), exec_info=None, **arguments)
    for ex, done in events:
        ex.finish(done)
    for shift, sub in strips:
        stencil.run(_domain_=tuple(sub), _origin_=_shifted(origin, shift), exec_info=None, **arguments)
