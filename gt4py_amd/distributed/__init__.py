"""Multi-GPU execution of a stencil: IJ decomposition + halo exchange overlapped with compute.

NEW relative to the reference (gt4py.cartesian is single-device).
"""

from __future__ import annotations

from typing import Any, Dict, Mapping, Sequence

from .halo import (Decomposition, HaloExchanger, HipPacker, choose_process_grid, exchange_cost, halo_boxes, halo_sides,
                   process_grid_candidates, receive_order, scatter_global)
from .native import NativeComm, NativeHaloExchanger
from .selfcheck import FormCheck, coordinate_values, count_wrong_cells

__all__ = ["Decomposition", "FormCheck", "HaloExchanger", "HipPacker", "NativeComm", "NativeHaloExchanger", "TunedApply",
           "choose_process_grid", "coordinate_values", "count_wrong_cells", "exchange_cost", "fused_apply", "halo_boxes", "halo_sides", "overlapped_apply",
           "process_grid_candidates", "receive_order", "scatter_global", "sequential_apply"]


def _shifted(origin: Mapping[str, Sequence[int]], shift: Sequence[int]) -> Dict[str, tuple]:
    return {name: tuple(int(o) + int(s) for o, s in zip(org, shift)) for name, org in origin.items()}


_FUSED_CACHE: Dict[Any, Any] = {}


def fused_apply(stencil, decomp: Decomposition, origin: Mapping[str, Sequence[int]], arguments: Dict[str, Any],
                exchange: Mapping[str, Any]) -> bool:
    """The distributed apply as ONE native call, where the library has it: a stencil bound to the hand-written horizontal
    diffusion (``gt4mi_dist_hdiff_*``, ghost depth 2) or 5-point (``gt4mi_dist_lap5_f64 / _f32``, ghost depth 1) kernels whose
    read field is exchanged through a ``NativeHaloExchanger``.  Pack, interior kernel next to the exchange, then one ring
    kernel for the points that read ghost cells.  Returns False when the combination is not covered (the caller falls
    back to the stencil-agnostic schedule of ``overlapped_apply``)."""
    binding = getattr(type(stencil), "_gt_binding_", None)
    if binding is None or len(exchange) != 1:
        return False
    (name, ex), = exchange.items()
    if not isinstance(ex, NativeHaloExchanger):
        return False
    roles = binding.roles
    read_role = {"hdiff": "in_field", "lap5": "inp"}.get(binding.family)
    if read_role is None or roles.get(read_role) != name:
        return False
    if len({tuple(origin[roles[r]]) for r in roles if roles[r] in origin}) != 1:
        return False  # the native step takes ONE origin for all fields
    org = tuple(origin[name])
    itemsize = binding.dtype.itemsize
    if ex.itemsize != itemsize:
        return False
    field_names = [roles[r] for r in roles if roles[r] in arguments and hasattr(arguments[roles[r]], "ptr")]
    scalars = tuple(float(arguments[roles[r]]) for r in roles if roles[r] in arguments and not hasattr(arguments[roles[r]], "ptr"))
    key = (id(stencil), id(ex), org, tuple(id(arguments[n]) for n in field_names), scalars)
    entry = _FUSED_CACHE.get(key)
    # (the cached call holds raw pointers, not the arrays: it is only used while every array it was bound to is alive and IS
    # the one passed now -- a recycled id() of a dead array fails the identity check and the call is rebuilt)
    if entry is None or any(r() is not arguments[n] for n, r in entry[1]):
        import weakref

        if binding.family == "hdiff":
            if decomp.halo != 2:
                return False
            coeff_name = roles["coeff"]
            coeff = arguments[coeff_name]
            is_field = hasattr(coeff, "ptr")
            call = ex.make_dist_hdiff(arguments[name], arguments[roles["out_field"]], coeff if is_field else None, org,
                                      binding.flags, 0.0 if is_field else float(coeff), hold_arrays=False)
        else:
            if decomp.halo != 1 or itemsize not in (4, 8) or (binding.flags and itemsize == 8):
                return False
            call = ex.make_dist_lap5(arguments[name], arguments[roles["out"]], org, org, binding.variant, hold_arrays=False,
                                     flags=binding.flags)
        if len(_FUSED_CACHE) >= 16:
            _FUSED_CACHE.pop(next(iter(_FUSED_CACHE)))
        entry = _FUSED_CACHE[key] = (call, [(n, weakref.ref(arguments[n])) for n in field_names])
    entry[0]()
    return True


def overlapped_apply(stencil, decomp: Decomposition, origin: Mapping[str, Sequence[int]], arguments: Dict[str, Any],
                     exchange: Mapping[str, HaloExchanger], fused: bool = True) -> None:
    """One distributed apply of ``stencil`` on this rank.

    ``exchange`` maps the names of the read fields whose ghost cells are refreshed this step to
    their exchangers.  Timeline (two HIP streams):

        side stream : pack -> RCCL send/recv (I faces, then J faces) -> unpack
        main stream : interior kernel  ................  wait  -> boundary-strip kernels

    Stencils bound to the kernel library's horizontal diffusion / 5-point kernels take the native fused step instead
    (``fused_apply``: one C call, one ring kernel for all boundary strips) unless ``fused=False``.
    The stencil must have been built with ``device_sync=False`` so that launches stay asynchronous.
    ``arguments`` holds the device arrays / scalars by parameter name; ``origin`` the per-field origin
    of the LOCAL compute domain.
    """
    if fused and fused_apply(stencil, decomp, origin, arguments, exchange):
        return
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
