"""Multi-GPU execution of a stencil: IJ decomposition + halo exchange overlapped with compute.

NEW relative to the reference (gt4py.cartesian is single-device).
"""

from __future__ import annotations

from typing import Any, Dict, Mapping, Sequence

from .halo import (Decomposition, HaloExchanger, HipPacker, choose_process_grid, exchange_cost, halo_boxes,
                   process_grid_candidates, scatter_global)
from .native import NativeComm, NativeHaloExchanger

__all__ = ["Decomposition", "HaloExchanger", "HipPacker", "NativeComm", "NativeHaloExchanger", "TunedApply",
           "choose_process_grid", "exchange_cost", "halo_boxes", "overlapped_apply", "process_grid_candidates",
           "scatter_global", "sequential_apply"]


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
    pending = []  # (exchanger, is_native, handle)
    native = [(name, ex) for name, ex in exchange.items() if isinstance(ex, NativeHaloExchanger)]
    for name, ex in exchange.items():
        if not isinstance(ex, NativeHaloExchanger):
            pending.append((ex, False, ex.start(arguments[name].tensor)))
    # The exchange is enqueued BEFORE the interior kernel: its pack and send/recv kernels get onto the device
    # while it is idle.  Once the interior kernel saturates HBM, the few workgroups of a send/recv kernel see
    # loaded-memory latency and crawl (measured: 13 us alone, 170 us next to a 185 us hdiff interior), so
    # whatever finishes before that is a gain (profiles/r1_dist_hdiff_rehearsal.log).
    for name, ex in native:  # pack + RCCL + unpack on the side stream inside one C call
        ex.fork()
        ex.begin(arguments[name])
        pending.append((ex, True, None))
    (shift, sub), strips = decomp.interior_and_strips()
    if all(d > 0 for d in sub):
        stencil.run(_domain_=tuple(sub), _origin_=_shifted(origin, shift), exec_info=None, **arguments)
    for ex, is_native, handle in pending:
        if is_native:
            ex.end()
        else:
            ex.finish(handle)
    for shift, sub in strips:
        stencil.run(_domain_=tuple(sub), _origin_=_shifted(origin, shift), exec_info=None, **arguments)


def sequential_apply(stencil, decomp: Decomposition, origin: Mapping[str, Sequence[int]], arguments: Dict[str, Any],
                     exchange: Mapping[str, Any]) -> None:
    """The same distributed apply without overlap: refresh the ghost cells, then ONE launch over the whole local
    domain, all on the caller's stream.  No boundary strips and nothing for the send/recv kernels to compete
    with -- faster than ``overlapped_apply`` whenever the exchange is short next to the strips it saves
    (profiles/r1_dist_hdiff_rehearsal.log)."""
    for name, ex in exchange.items():
        if isinstance(ex, NativeHaloExchanger):
            ex.exchange(arguments[name])
        else:
            ex.finish(ex.start(arguments[name].tensor))
    stencil.run(_domain_=tuple(decomp.local_domain), _origin_={n: tuple(o) for n, o in origin.items()}, exec_info=None,
                **arguments)


class TunedApply:
    """Distributed apply that measures both forms once and keeps the faster one.

    Which of ``overlapped_apply`` / ``sequential_apply`` wins depends on the stencil (how expensive its strips
    are), the local domain and the links, none of which is known up front.  ``calibrate()`` times a few
    applies of each on the current stream; with an initialised ``torch.distributed`` group every rank adopts
    the choice that is best for the slowest rank.  The timed applies run on CLONES of every field the stencil
    writes, so the caller's data is not touched (an in/out field of a time-stepping stencil would otherwise
    advance by the number of calibration applies); read-only fields are used in place, and their ghost cells
    are refreshed by the exchanges, which an ordinary apply does as well."""

    def __init__(self, stencil, decomp: Decomposition, origin: Mapping[str, Sequence[int]], exchange: Mapping[str, Any]):
        self.stencil, self.decomp, self.origin, self.exchange = stencil, decomp, origin, exchange
        self.choice = None
        self.timings_ms: Dict[str, float] = {}

    def _scratch_arguments(self, arguments: Dict[str, Any]) -> Dict[str, Any]:
        from ..cartesian.definitions import AccessKind

        out = dict(arguments)
        info = getattr(self.stencil, "field_info", {}) or {}
        for name, fi in info.items():
            if name in arguments and fi is not None and fi.access in (AccessKind.WRITE, AccessKind.READ_WRITE):
                if name in self.exchange:
                    raise ValueError(f"field '{name}' is written by the stencil AND exchanged: calibrate() cannot keep it "
                                     "untouched; choose the form explicitly (TunedApply.choice = ...)")
                out[name] = arguments[name].copy() if hasattr(arguments[name], "copy") else arguments[name].clone()
        return out

    def calibrate(self, arguments: Dict[str, Any], iters: int = 8) -> str:
        import time

        import torch

        group = torch.distributed.is_available() and torch.distributed.is_initialized()
        scratch = self._scratch_arguments(arguments)
        forms = {"overlapped": overlapped_apply, "sequential": sequential_apply}
        for name, fn in forms.items():
            for _ in range(2):
                fn(self.stencil, self.decomp, self.origin, scratch, self.exchange)
            torch.cuda.synchronize()
            if group:
                torch.distributed.barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn(self.stencil, self.decomp, self.origin, scratch, self.exchange)
            torch.cuda.synchronize()
            dt = torch.tensor([(time.perf_counter() - t0) / iters * 1e3], dtype=torch.float64, device="cuda")
            if group:
                torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
            self.timings_ms[name] = float(dt.item())
        self.choice = min(self.timings_ms, key=self.timings_ms.get)
        return self.choice

    def __call__(self, arguments: Dict[str, Any]) -> None:
        if self.choice is None:
            self.calibrate(arguments)
        fn = overlapped_apply if self.choice == "overlapped" else sequential_apply
        fn(self.stencil, self.decomp, self.origin, arguments, self.exchange)
