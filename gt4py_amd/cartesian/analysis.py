"""Static analysis of an ``ir.Stencil``: horizontal extents, access kinds, K boundaries, minimum K
size -> the ``field_info`` / ``parameter_info`` / ``domain_info`` a StencilObject exposes.

Restates, for the small IR, the rules of
* ``StencilExtentComputer`` (/root/reference/src/gt4py/cartesian/gtc/passes/oir_optimizations/utils.py:250-313):
  statements are visited last to first; the block extent of a statement is the union of the extents
  already required of the fields it writes (zero for fields nobody reads later); every read then
  requires ``block extent + offset`` of the field read;
* ``AccessKindComputer`` (gtc/passes/oir_access_kinds.py:21-67): right-hand side before left;
  WRITE after READ = READ_WRITE, READ after WRITE stays WRITE;
* ``compute_k_boundary`` / ``compute_min_k_size`` (gtc/passes/gtir_k_boundary.py:24-109);
* ``make_args_data_from_gtir`` (backend/module_generator.py:56-106).
Pinned by SURVEY.md Appendix E.1 and the reference tests cited there.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from . import ir
from .definitions import AccessKind, Boundary, DomainInfo, FieldInfo, ParameterInfo

Extent2 = Tuple[Tuple[int, int], Tuple[int, int]]  # ((i_lo, i_hi), (j_lo, j_hi)), lo <= 0 <= hi
ZERO_EXTENT: Extent2 = ((0, 0), (0, 0))


def _union(a: Extent2, b: Extent2) -> Extent2:
    return tuple((min(x[0], y[0]), max(x[1], y[1])) for x, y in zip(a, b))  # type: ignore[return-value]


def _shift(block: Extent2, offset: Tuple[int, int, int]) -> Extent2:
    return tuple((lo + o, hi + o) for (lo, hi), o in zip(block, offset[:2]))  # type: ignore[return-value]


@dataclass
class ExtentInfo:
    fields: Dict[str, Extent2]
    blocks: List[Extent2]  # one per statement, in program order


def access_extent(block: Extent2, offset: Tuple[int, int, int], region: "ir.Region | None" = None):
    """Extent a read at ``offset`` requires of its field when the statement runs on ``block``; None when
    a horizontal mask does not overlap the block (GenericAccess.to_extent,
    gtc/passes/oir_optimizations/utils.py:52-76)."""
    if region is None:
        return _shift(block, offset)
    di, dj = region.i.overlap(block[0]), region.j.overlap(block[1])
    if di is None or dj is None:
        return None
    reach = ((block[0][0] - di[0], block[0][1] - di[1]), (block[1][0] - dj[0], block[1][1] - dj[1]))
    return _union(_shift(reach, offset), ZERO_EXTENT)


def compute_extents(stencil: ir.Stencil) -> ExtentInfo:
    stmts = [s for _, _, s in stencil.statements()]
    fields: Dict[str, Extent2] = {}
    blocks: List[Extent2] = [ZERO_EXTENT] * len(stmts)
    idx = len(stmts) - 1
    while idx >= 0:
        # one horizontal execution = one plain statement, or all statements flattened from one `if`
        first = idx
        if stmts[idx].group >= 0:
            while first > 0 and stmts[first - 1].group == stmts[idx].group:
                first -= 1
        members = range(first, idx + 1)
        block = ZERO_EXTENT
        for m in members:
            block = _union(block, fields.setdefault(stmts[m].target.name, ZERO_EXTENT))
        for m in members:
            blocks[m] = block
        for m in members:
            for e in ir.stmt_reads(stmts[m]):
                if isinstance(e, ir.FieldAccess):
                    need = access_extent(block, e.offset, stmts[m].region)
                    if need is not None:
                        fields[e.name] = _union(fields[e.name], need) if e.name in fields else need
        idx = first - 1
    for f in stencil.fields:
        fields.setdefault(f.name, ZERO_EXTENT)
    return ExtentInfo(fields, blocks)


def storage_extents(stencil: ir.Stencil, extents: ExtentInfo) -> Dict[str, Extent2]:
    """Extent a temporary's storage must cover: everything read of it AND every block it is written on.

    The two differ when a temporary is only ever read at non-zero offsets (read extent ((-1, -1), ...),
    written on ((-1, 0), ...)): sizing the array by the read extent alone, as the reference's numpy
    backend does (gtc/numpy/oir_to_npir.py:42-56), leaves the write one element short."""
    out: Dict[str, Extent2] = {t.name: extents.fields.get(t.name, ZERO_EXTENT) for t in stencil.temporaries}
    for (_, _, stmt), block in zip(stencil.statements(), extents.blocks):
        if stmt.target.name in out:
            out[stmt.target.name] = _union(out[stmt.target.name], block)
    return out


def compute_access_kinds(stencil: ir.Stencil) -> Dict[str, AccessKind]:
    access: Dict[str, AccessKind] = {}

    def touch(name: str, kind: AccessKind) -> None:
        if kind == AccessKind.WRITE and access.get(name) == AccessKind.READ:
            access[name] = AccessKind.READ_WRITE
        elif name not in access:
            access[name] = kind

    for _, _, stmt in stencil.statements():
        for e in ir.stmt_reads(stmt):
            if isinstance(e, (ir.FieldAccess, ir.ScalarAccess)):
                touch(e.name, AccessKind.READ)
        touch(stmt.target.name, AccessKind.WRITE)
    return access


class TemporaryReadOutsideDomain(NotImplementedError):
    """A temporary is read beyond the K range of the domain from a vertical loop other than the one that declares it.

    The reference's GTIR accepts this (gtir_k_boundary.py:64-67 checks only temporaries declared in the current vertical
    loop) and its numpy backend then fails at run time on the slice; here scratch arrays hold exactly the domain's
    levels, so the stencil is refused at build time -- a divergence from the reference, hence not its ``TypeError``."""


def compute_k_boundary(stencil: ir.Stencil) -> Dict[str, Tuple[int, int]]:
    """K boundary per field; ``TypeError`` for a temporary accessed outside the K range it exists on.

    gtir_k_boundary.py:39-70 raises for the vertical loop that declares the temporary (= first assigns it,
    defir_to_gtir.py:433-451): the reference's ``TypeError`` and message.  A temporary read beyond the domain from a
    LATER loop is not caught there but fails in the numpy backend at run time (temporaries hold exactly ``_dK_``
    levels, npir_codegen.py:88-104); scratch arrays here hold exactly the domain's levels too, so that form is
    rejected as well, with ``TemporaryReadOutsideDomain`` (not the reference's error, because the
    reference accepts the program)."""
    neg_inf = float("-inf")
    bounds: Dict[str, Tuple[float, float]] = {d.name: (neg_inf, neg_inf) for d in (*stencil.fields, *stencil.temporaries)}
    temporaries = {t.name for t in stencil.temporaries}
    declared_in: Dict[str, int] = {}  # temporary -> id of the vertical loop (computation) that first assigns it
    for comp, block, stmt in stencil.statements():
        if stmt.target.name in temporaries:
            declared_in.setdefault(stmt.target.name, id(comp))
        accesses = [stmt.target] + [e for e in ir.stmt_reads(stmt) if isinstance(e, ir.FieldAccess)]
        for acc in accesses:
            lo, hi = bounds[acc.name]
            if acc.koffset is None:  # gtir_k_boundary.py:52: variable offsets do not bound anything
                if block.interval.start.level is ir.Level.START:
                    lo = max(-block.interval.start.offset - acc.offset[2], lo)
                if block.interval.end.level is ir.Level.END:
                    hi = max(block.interval.end.offset + acc.offset[2], hi)
            if acc.name in temporaries and (lo > 0 or hi > 0):
                if declared_in.get(acc.name, id(comp)) == id(comp):
                    raise TypeError(f"Invalid access with offset in k to temporary field {acc.name}.")
                raise TemporaryReadOutsideDomain(
                    f"temporary field {acc.name} is read beyond the K range of the domain from a later vertical loop: the "
                    f"reference accepts this at build time (gtir_k_boundary.py:64-67) and fails in its numpy backend at run "
                    f"time; backend temporaries here hold exactly the domain's levels")
            bounds[acc.name] = (lo, hi)
    return {n: (int(lo) if lo != neg_inf else 0, int(hi) if hi != neg_inf else 0) for n, (lo, hi) in bounds.items()}


def compute_min_k_size(stencil: ir.Stencil) -> int:
    min_start = min_end = biggest = 0
    for comp in stencil.computations:
        for block in comp.blocks:
            s, e = block.interval.start, block.interval.end
            if s.level is ir.Level.START and e.level is ir.Level.END:
                if not (s.offset == 0 and e.offset == 0):
                    biggest = max(biggest, s.offset - e.offset + 1)
            elif s.level is ir.Level.START and e.level is ir.Level.START:
                min_start = max(min_start, e.offset)
                biggest = max(biggest, e.offset)
            else:
                min_end = max(min_end, -s.offset)
                biggest = max(biggest, -s.offset)
    return max(min_start + min_end, biggest)


@dataclass
class ArgsData:
    field_info: Dict[str, FieldInfo]
    parameter_info: Dict[str, ParameterInfo]
    domain_info: DomainInfo
    extents: ExtentInfo


def make_args_data(stencil: ir.Stencil) -> ArgsData:
    extents = compute_extents(stencil)
    access = compute_access_kinds(stencil)
    k_bounds = compute_k_boundary(stencil)
    field_info: Dict[str, FieldInfo] = {}
    for decl in stencil.fields:
        kind = access.get(decl.name, AccessKind.NONE)
        if kind != AccessKind.NONE:
            (ilo, ihi), (jlo, jhi) = extents.fields[decl.name]
            klo, khi = k_bounds[decl.name]
            # extent -> boundary: (-lo, hi), NOT clamped at zero (gtc/definitions.py:565-566 Extent.to_boundary on the
            # un-centred field extents): a field only read at [1, 0, 0] has boundary (-1, 1) in I, which is what
            # allows origin -1 for it (test_code_generation.py:520-558)
            boundary = Boundary(((-ilo, ihi), (-jlo, jhi), (klo, khi)))
        else:
            boundary = Boundary.zeros(3)
        field_info[decl.name] = FieldInfo(access=kind, boundary=boundary, axes=tuple(decl.axes),
                                          data_dims=tuple(decl.data_dims), dtype=np.dtype(decl.dtype))
    parameter_info = {
        p.name: ParameterInfo(access=access.get(p.name, AccessKind.NONE), dtype=np.dtype(p.dtype))
        for p in stencil.params
    }
    domain_info = DomainInfo(parallel_axes=("I", "J"), sequential_axis="K",
                             min_sequential_axis_size=compute_min_k_size(stencil), ndim=3)
    return ArgsData(field_info, parameter_info, domain_info, extents)
