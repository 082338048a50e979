"""Generic executor, planning half: typed stencil IR -> inlined IR -> stages -> storage classes.

Passes 1-3 of the pipeline described in ``hip_codegen.py`` (which emits HIP source from the ``Plan``
built here):

1. ``inline_horizontal_temporaries`` -- recompute instead of staging (the reference's on-the-fly merging,
   gtc/passes/oir_optimizations/horizontal_execution_merging.py:135-330);
2. ``plan_stages`` -- cut program order into kernels wherever one column reads what another wrote, choose
   thread-per-point or thread-per-column per stage;
3. storage classes of the temporaries (thread-local / scratch / register-forwarded / register-only).
"""

from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Set, Tuple

import numpy as np

from .. import analysis, ir

Extent2 = analysis.Extent2


class UnsupportedStencil(NotImplementedError):
    """The stencil is outside what the generic executor can run exactly."""


# ---------------------------------------------------------------------------------------------------
# 1. inlining of horizontally offset temporaries
# ---------------------------------------------------------------------------------------------------
def _shift_access(e: ir.FieldAccess, shift: Tuple[int, int]) -> ir.FieldAccess:
    return replace(e, offset=(e.offset[0] + shift[0], e.offset[1] + shift[1], e.offset[2]))


#: substitute temporaries that are assigned under a run-time `if` as selects (GT4MI_PLAN_INLINE_MASKED=0: keep them in
#: memory as before)
INLINE_MASKED = __import__("os").environ.get("GT4MI_PLAN_INLINE_MASKED", "1") != "0"


def merge_if_else_assignments(stencil: ir.Stencil) -> ir.Stencil:
    """`if c: x = a` ... `else: x = b` arrives as two conditional assignments, `x = where(c, a, x)` and
    `x = where(not c, b, x)`.  When nothing in between touches `x`, `c` or what `a` reads, the pair IS
    `x = a if c else b`: one plain assignment (stored as whole vectors with streaming stores by the strip kernels, a full
    definition for the passes below) instead of two stores point by point under a condition."""
    if __import__("os").environ.get("GT4MI_PLAN_MERGE_IF_ELSE", "1") == "0":
        return stencil
    changed = False
    new_comps = []
    for comp in stencil.computations:
        new_blocks = []
        for block in comp.blocks:
            body = list(block.body)
            i = 0
            while i < len(body):
                first = body[i]
                t = first.target
                ok = (first.mask is not None and first.region is None and not first.loops and tuple(t.offset) == (0, 0, 0)
                      and t.koffset is None and not t.data_index)
                if ok:
                    needs = {e.name for e in ir.stmt_reads(first) if isinstance(e, ir.FieldAccess)}  # of the value and of c
                    for j in range(i + 1, len(body)):
                        other = body[j]
                        if other.target.name == t.name:
                            if (other.region is None and not other.loops and other.target == t and other.mask is not None
                                    and other.mask == ir.UnaryOp("not", first.mask, other.mask.dtype)
                                    and not any(isinstance(e, ir.FieldAccess) and e.name == t.name for e in ir.walk(other.value))):
                                dt = np.dtype(stencil.decl(t.name).dtype)
                                body[j] = ir.Assign(t, ir.TernaryOp(first.mask, first.value, other.value, dt), None, other.group,
                                                    None, ())
                                del body[i]
                                changed = True
                                i -= 1
                            break
                        reads = {e.name for e in ir.stmt_reads(other) if isinstance(e, ir.FieldAccess)}
                        if t.name in reads or other.target.name in needs:
                            break
                i += 1
            new_blocks.append(ir.IntervalBlock(block.interval, tuple(body)))
        new_comps.append(ir.Computation(comp.order, tuple(new_blocks)))
    if not changed:
        return stencil
    return ir.Stencil(stencil.name, stencil.fields, stencil.params, stencil.temporaries, tuple(new_comps))


def split_block_local_temporaries(stencil: ir.Stencil) -> ir.Stencil:
    """A temporary that several interval blocks use, each for itself -- every block that touches it assigns it (plainly:
    no mask, region or loop) before it reads it, and nobody reads it at a K offset -- is one temporary per block under one
    name: no value flows between the blocks (at a level, a block only ever sees what it assigned itself).  Give every
    block its own name, so that the passes below treat them as what they are (inlined or thread-local per block instead
    of one scratch array that forces a stage per statement).  Typical: the same `lap` / `flx` names in the boundary
    interval and in the interior interval of a horizontal operator."""
    if __import__("os").environ.get("GT4MI_PLAN_SPLIT_BLOCK_LOCAL", "1") == "0":
        return stencil
    temps = {t.name: t for t in stencil.temporaries}
    first_is_write: Dict[str, Dict[Tuple[int, int], bool]] = {}
    shared_between_blocks: Set[str] = set()
    for ci, comp in enumerate(stencil.computations):
        for bi, block in enumerate(comp.blocks):
            seen: Set[str] = set()
            partial: Dict[str, ir.Expr] = {}  # assigned in the first branch of an `if` so far: name -> its condition
            for stmt in block.body:
                for e in ir.stmt_reads(stmt):
                    if isinstance(e, ir.FieldAccess) and e.name in temps:
                        if e.offset[2] != 0 or e.koffset is not None:
                            shared_between_blocks.add(e.name)
                        if e.name not in seen or e.name in partial:  # read before the block has assigned it everywhere
                            seen.add(e.name)
                            partial.pop(e.name, None)
                            first_is_write.setdefault(e.name, {})[(ci, bi)] = False
                t = stmt.target
                # a statement between the two branches that writes what the condition reads: `not c` is no longer the
                # complement of the `c` the first branch saw (merge_if_else_assignments checks the same through `needs`)
                for name in [n for n, c in partial.items() if n != t.name
                             and any(isinstance(e, ir.FieldAccess) and e.name == t.name for e in ir.walk(c))]:
                    del partial[name]  # first_is_write stays False
                if t.name not in temps:
                    continue
                simple = (stmt.region is None and not stmt.loops and tuple(t.offset) == (0, 0, 0) and t.koffset is None
                          and not t.data_index)
                if t.name in partial:  # the `else` branch completes the assignment; anything else leaves it partial
                    cond = partial.pop(t.name)
                    complete = simple and stmt.mask is not None and stmt.mask == ir.UnaryOp("not", cond, stmt.mask.dtype)
                    first_is_write[t.name][(ci, bi)] = complete
                elif t.name not in seen:
                    seen.add(t.name)
                    if simple and stmt.mask is not None:
                        partial[t.name] = stmt.mask  # decided by what follows
                        first_is_write.setdefault(t.name, {})[(ci, bi)] = False
                    else:
                        first_is_write.setdefault(t.name, {})[(ci, bi)] = simple and stmt.mask is None
    split = {n for n, blocks in first_is_write.items()
             if len(blocks) > 1 and all(blocks.values()) and n not in shared_between_blocks and not temps[n].data_dims
             and tuple(temps[n].axes) == ("I", "J", "K")}
    if not split:
        return stencil
    new_temps: List[ir.FieldDecl] = [t for t in stencil.temporaries if t.name not in split]
    new_comps = []
    for ci, comp in enumerate(stencil.computations):
        new_blocks = []
        for bi, block in enumerate(comp.blocks):
            names = {n: f"{n}__b{ci}_{bi}" for n in split if (ci, bi) in first_is_write[n]}
            if not names:
                new_blocks.append(block)
                continue
            new_temps.extend(replace(temps[n], name=new) for n, new in names.items())

            def fn(e, names=names):
                if isinstance(e, ir.FieldAccess) and e.name in names:
                    return replace(e, name=names[e.name])
                return e

            body = []
            for stmt in block.body:
                body.append(ir.Assign(ir.map_expr(stmt.target, fn), ir.map_expr(stmt.value, fn),
                                      ir.map_expr(stmt.mask, fn) if stmt.mask is not None else None, stmt.group, stmt.region,
                                      tuple((lid, ir.map_expr(c, fn)) for lid, c in stmt.loops)))
            new_blocks.append(ir.IntervalBlock(block.interval, tuple(body)))
        new_comps.append(ir.Computation(comp.order, tuple(new_blocks)))
    return ir.Stencil(stencil.name, stencil.fields, stencil.params, tuple(new_temps), tuple(new_comps))


def inline_horizontal_temporaries(stencil: ir.Stencil) -> Tuple[ir.Stencil, Set[str]]:
    """Returns (rewritten stencil, names of SSA values that are thread-local by construction)."""
    new_stencil, ssa_locals, _ = inline_horizontal_temporaries_with_forms(stencil)
    return new_stencil, ssa_locals


def inline_horizontal_temporaries_with_forms(stencil: ir.Stencil):
    """As ``inline_horizontal_temporaries``, plus {(computation, block): (order, defs)}: the single-assignment form of
    every interval block whose temporaries were inlined -- ``order`` lists ("def", version name) and ("stmt", Assign) in
    program order, ``defs`` maps a version name to its defining expression over fields and earlier versions (offsets
    as written).  The strip kernels with shared temporaries (hip_codegen) are generated from this form."""
    stencil = split_block_local_temporaries(merge_if_else_assignments(stencil))
    written = {s.target.name for _, _, s in stencil.statements()}
    pure_inputs = {f.name for f in stencil.fields if f.name not in written}
    temps = {t.name: t for t in stencil.temporaries}

    touched: Dict[str, Set[Tuple[int, int]]] = {}
    k_offset_read: Set[str] = set()
    ij_offset_read: Set[str] = set()
    masked_write: Set[str] = set()  # conditionally assigned: the old value shows through, no substitution
    for ci, comp in enumerate(stencil.computations):
        for bi, block in enumerate(comp.blocks):
            for stmt in block.body:
                touched.setdefault(stmt.target.name, set()).add((ci, bi))
                # a run-time `if` (mask only) is a select against the value assigned before -- it can be substituted like
                # any other definition; inside a horizontal region or a `while` the old value shows through by position
                # or by iteration: no substitution
                if stmt.region is not None or stmt.loops or (stmt.mask is not None and not INLINE_MASKED):
                    masked_write.add(stmt.target.name)
                for e in ir.stmt_reads(stmt):
                    if isinstance(e, ir.FieldAccess):
                        touched.setdefault(e.name, set()).add((ci, bi))
                        if e.offset[2] != 0 or e.koffset is not None:
                            k_offset_read.add(e.name)
                        if e.offset[0] != 0 or e.offset[1] != 0:
                            ij_offset_read.add(e.name)

    cand = {
        n for n in temps
        if len(touched.get(n, ())) == 1 and n not in k_offset_read and n not in masked_write and not temps[n].data_dims
        and tuple(temps[n].axes) == ("I", "J", "K")
        and stencil.computations[next(iter(touched[n]))[0]].order is ir.LoopOrder.PARALLEL
    }
    changed = True
    while changed:
        changed = False
        for _, _, stmt in stencil.statements():
            if stmt.target.name in cand:
                for e in ir.stmt_reads(stmt):
                    if isinstance(e, ir.FieldAccess) and e.name not in pure_inputs and e.name not in cand:
                        cand.discard(stmt.target.name)
                        changed = True
                        break
    # inline what is read at a horizontal offset, plus everything such a definition refers to
    inline = {n for n in cand if n in ij_offset_read}
    changed = True
    while changed:
        changed = False
        for _, _, stmt in stencil.statements():
            if stmt.target.name in inline:
                for e in ir.stmt_reads(stmt):
                    if isinstance(e, ir.FieldAccess) and e.name in cand and e.name not in inline:
                        inline.add(e.name)
                        changed = True
    if not inline:
        return stencil, set(), {}

    forms: Dict[Tuple[int, int], Tuple[List[Tuple[str, object]], Dict[str, ir.Expr]]] = {}
    ssa_locals: Dict[str, np.dtype] = {}
    new_comps = []
    for ci, comp in enumerate(stencil.computations):
        new_blocks = []
        for bi, block in enumerate(comp.blocks):
            version: Dict[str, str] = {}
            defs: Dict[str, ir.Expr] = {}
            order: List[Tuple[str, object]] = []  # ("def", version) | ("stmt", Assign)

            def to_versions(expr: ir.Expr) -> ir.Expr:
                def fn(e):
                    if isinstance(e, ir.FieldAccess) and e.name in inline:
                        if e.name not in version:  # the frontend rejects this already
                            raise UnsupportedStencil(f"temporary '{e.name}' is read before it is assigned")
                        return replace(e, name=version[e.name])
                    return e

                return ir.map_expr(expr, fn)

            for stmt in block.body:
                value = to_versions(stmt.value)
                mask = to_versions(stmt.mask) if stmt.mask is not None else None
                loops = tuple((lid, to_versions(c)) for lid, c in stmt.loops)
                name = stmt.target.name
                if name in inline:
                    v = f"{name}__v{sum(1 for k in defs if k.rsplit('__v', 1)[0] == name)}"
                    if mask is not None:
                        # target = where(mask, value, target): a select against the version assigned so far (a temporary
                        # that was not assigned before holds nothing defined -- the reference allocates it with
                        # Field.empty, gtc/numpy/npir_codegen.py:104-107 --: 0 stands in)
                        dt = np.dtype(temps[name].dtype)
                        old = (ir.FieldAccess(version[name], (0, 0, 0), dt) if name in version
                               else ir.Literal(False if dt == np.dtype("bool") else dt.type(0).item(), dt))
                        value = ir.TernaryOp(mask, value, old, dt)
                    defs[v] = value
                    version[name] = v
                    order.append(("def", v))
                else:
                    order.append(("stmt", ir.Assign(to_versions(stmt.target), value, mask, stmt.group, stmt.region, loops)))

            memo: Dict[Tuple[str, Tuple[int, int]], ir.Expr] = {}
            needed: Set[str] = set()

            def expand(expr: ir.Expr, shift: Tuple[int, int]) -> ir.Expr:
                def fn(e):
                    if isinstance(e, ir.FieldAccess):
                        if e.name in defs:
                            s = (shift[0] + e.offset[0], shift[1] + e.offset[1])
                            if s == (0, 0):
                                needed.add(e.name)
                                return ir.FieldAccess(e.name, (0, 0, 0), e.dtype)
                            key = (e.name, s)
                            if key not in memo:
                                memo[key] = expand(defs[e.name], s)
                            return memo[key]
                        return _shift_access(e, shift) if shift != (0, 0) else e
                    return e

                return ir.map_expr(expr, fn)

            body_stmts = [(kind, (expand(obj.value, (0, 0)), expand(obj.mask, (0, 0)) if obj.mask is not None else None,
                                  tuple((lid, expand(c, (0, 0))) for lid, c in obj.loops))
                           if kind == "stmt" else None, obj) for kind, obj in order]
            local_defs: Dict[str, ir.Expr] = {}
            pending = list(needed)
            while pending:  # thread-local values referenced at the thread's own point
                v = pending.pop()
                if v in local_defs:
                    continue
                before = set(needed)
                local_defs[v] = expand(defs[v], (0, 0))
                pending.extend(needed - before)
            new_body = []
            for kind, value, obj in body_stmts:
                if kind == "def":
                    if obj in local_defs:
                        dt = local_defs[obj].dtype
                        ssa_locals[obj] = np.dtype(temps[obj.rsplit("__v", 1)[0]].dtype)
                        new_body.append(ir.Assign(ir.FieldAccess(obj, (0, 0, 0), ssa_locals[obj]), local_defs[obj]))
                else:
                    new_body.append(ir.Assign(expand(obj.target, (0, 0)), value[0], value[1], obj.group, obj.region, value[2]))
            new_blocks.append(ir.IntervalBlock(block.interval, tuple(new_body)))
            if defs:
                forms[(ci, bi)] = (order, defs)
        new_comps.append(ir.Computation(comp.order, tuple(new_blocks)))
    new_temps = tuple(t for t in stencil.temporaries if t.name not in inline) + tuple(
        ir.FieldDecl(n, dt, ("I", "J", "K"), (), False) for n, dt in ssa_locals.items())
    return ir.Stencil(stencil.name, stencil.fields, stencil.params, new_temps, tuple(new_comps)), set(ssa_locals), forms


# ---------------------------------------------------------------------------------------------------
# 2./3. stages and storage classes
# ---------------------------------------------------------------------------------------------------
@dataclass
class Stmt:
    target: ir.FieldAccess
    value: ir.Expr
    extent: Extent2
    mask: Optional[ir.Expr] = None
    region: Optional[ir.Region] = None
    loops: Tuple[Tuple[int, ir.Expr], ...] = ()


@dataclass
class Nest:
    order: ir.LoopOrder
    interval: ir.Interval
    stmts: List[Stmt]
    block_id: Tuple[int, int]
    split_statements: bool = False  # PARALLEL block that has to run statement by statement over K


@dataclass
class Stage:
    nests: List[Nest] = field(default_factory=list)
    written: Set[str] = field(default_factory=set)
    offset_reads: Set[str] = field(default_factory=set)
    mapping: str = "ijk"
    extent: Extent2 = analysis.ZERO_EXTENT
    #: stages of a sequential interval block whose columns depend on each other run PLANE BY PLANE: the host
    #: loops over the K levels in sweep order and launches the block's stages once per level.
    #: (loop id, order, interval); consecutive stages with the same loop id are one such loop's body.
    plane: Optional[Tuple[int, ir.LoopOrder, ir.Interval]] = None


@dataclass
class TopCache:
    """Fields of a two-sweep column stage whose TOP levels can stay on chip between the sweeps.

    A FORWARD sweep up to the last level followed by a BACKWARD sweep from it (a Thomas solve) writes a value per
    level and reads it back at the same level on the way down; the levels written last are read first.  For
    ``names`` every access of the stage is at the thread's own column and level (or the level behind, through the
    forwarded register), so the last levels can live in registers and LDS instead of making a round trip through
    HBM -- what tridiag_stack.hip.h does by hand.  ``store_through``: fields that must be in memory afterwards
    anyway (API fields, temporaries another stage reads); for the others the cached levels are never stored."""

    names: Tuple[str, ...]
    store_through: Set[str]
    first_sweep_nests: int  # nests [0, n) are the FORWARD sweep, the rest the BACKWARD one
    start_margin: int  # largest START-relative interval bound: the cached range must begin above it


@dataclass
class Plan:
    stencil: ir.Stencil  # after inlining
    stages: List[Stage]
    field_extents: Dict[str, Extent2]
    locals: Set[str]  # thread-local temporaries
    scratch: Dict[str, Tuple[np.dtype, Extent2]]  # temporaries in global memory
    forwarded: Dict[Tuple[int, str], int]  # (stage index, name) -> +-1: value of the previous level kept in a register
    prime: Dict[Tuple[int, int, str], Tuple[str, Optional[int]]]  # (stage, nest, name) -> (mode, previous nest)
    register_only: Set[str]  # forwarded temporaries that never need memory
    api_fields: List[ir.FieldDecl]  # API fields a kernel touches
    params: List[ir.ScalarDecl]  # scalar parameters a kernel reads
    top_cache: Dict[int, TopCache] = field(default_factory=dict)  # stage index -> what may stay on chip
    #: (computation, block) -> single-assignment form of a block whose temporaries were inlined (see
    #: inline_horizontal_temporaries_with_forms)
    shared_forms: Dict[Tuple[int, int], Tuple[list, dict]] = field(default_factory=dict)


def _field_reads(expr: ir.Expr):
    return [e for e in ir.walk(expr) if isinstance(e, ir.FieldAccess)]


def _stmt_field_reads(s) -> List[ir.FieldAccess]:
    """Field reads of a planned statement or an ``ir.Assign``: mask and value."""
    reads = _field_reads(s.value)
    if s.mask is not None:
        reads = _field_reads(s.mask) + reads
    for _, cond in getattr(s, "loops", ()):
        reads = _field_reads(cond) + reads
    for sub in _target_exprs(s.target):  # run-time K offset / data index of the write
        reads = reads + _field_reads(sub)
    return reads


def _target_exprs(target: ir.FieldAccess) -> List[ir.Expr]:
    out = [target.koffset] if target.koffset is not None else []
    return out + [d for d in target.data_index or () if isinstance(d, ir.Expr)]


def _loop_invariant_statements(stmts: List[Stmt], temporaries: Set[str]) -> Tuple[List[Stmt], List[Stmt]]:
    """Split a sequential block's statements into (those that can run before the loop, the rest).

    A statement qualifies when it is an unconditional, unrestricted assignment (no mask, region or `while`) to a
    3-d temporary at the current level that is the ONLY write of that temporary in the block, reads nothing the
    block writes except temporaries that qualified before it, and whose temporary is read in the block only after
    it, at K offset 0: then the value of every level is the same whether it is computed inside the sweep or up
    front."""
    writes_of: Dict[str, int] = {}
    for s in stmts:
        writes_of[s.target.name] = writes_of.get(s.target.name, 0) + 1
    hoisted: List[Stmt] = []
    names: Set[str] = set()
    changed = True
    while changed:
        changed = False
        for idx, s in enumerate(stmts):
            t = s.target.name
            if any(s is h for h in hoisted) or t not in temporaries or writes_of[t] != 1:
                continue
            if s.mask is not None or s.region is not None or s.loops or s.target.offset != (0, 0, 0) \
                    or s.target.koffset is not None or s.target.data_index:
                continue
            reads = _stmt_field_reads(s)
            if any(e.name in writes_of and e.name not in names for e in reads):
                continue  # depends on the sweep
            if any(e.name in names and (e.offset[2] != 0 or e.koffset is not None) for e in reads):
                continue
            ok = True
            for jdx, other in enumerate(stmts):
                for e in _stmt_field_reads(other):
                    if e.name == t and (jdx <= idx or e.offset[2] != 0 or e.koffset is not None):
                        ok = False
            if ok:
                hoisted.append(s)
                names.add(t)
                changed = True
    hoisted.sort(key=lambda s: next(i for i, x in enumerate(stmts) if x is s))
    return hoisted, [s for s in stmts if not any(s is h for h in hoisted)]


def plan_stages(stencil_in: ir.Stencil) -> Plan:
    stencil, ssa_locals, shared_forms = inline_horizontal_temporaries_with_forms(stencil_in)
    extents = analysis.compute_extents(stencil)
    written_anywhere = {s.target.name for _, _, s in stencil.statements()}

    stages: List[Stage] = []
    cur: Optional[Stage] = None
    ext_iter = iter(extents.blocks)
    for ci, comp in enumerate(stencil.computations):
        for bi, block in enumerate(comp.blocks):
            stmts = [Stmt(s.target, s.value, next(ext_iter), s.mask, s.region, s.loops) for s in block.body]
            def per_statement(stmts):
                units = []  # one statement each, except that the body of a `while` stays together
                for st in stmts:
                    if st.loops and units and units[-1][-1].loops and units[-1][-1].loops[0][0] == st.loops[0][0]:
                        units[-1].append(st)
                    else:
                        units.append([st])
                return units

            def add(units, order, block_id, plane):
                nonlocal cur
                for unit in units:
                    if not unit:
                        continue
                    writes = {s.target.name for s in unit}
                    offreads = {
                        e.name for s in unit for e in _stmt_field_reads(s)
                        if (e.offset[0] != 0 or e.offset[1] != 0) and e.name in written_anywhere
                    }
                    if writes & offreads:
                        raise UnsupportedStencil(
                            f"{sorted(writes & offreads)} written and read at a horizontal offset inside one statement "
                            f"(or one `while` body) of a {comp.order.value} interval block")
                    if cur is not None and ((offreads & cur.written) or (writes & cur.offset_reads)):
                        cur = None
                    if cur is None:
                        cur = Stage(plane=plane)
                        stages.append(cur)
                    if cur.nests and cur.nests[-1].block_id == block_id:
                        cur.nests[-1].stmts.extend(unit)
                    else:
                        cur.nests.append(Nest(order, block.interval, list(unit), block_id))
                    cur.written |= writes
                    cur.offset_reads |= offreads

            def cross_column(body) -> bool:
                writes = {s.target.name for s in body}
                return bool(writes & {e.name for s in body for e in _stmt_field_reads(s) if e.offset[0] != 0 or e.offset[1] != 0})

            if comp.order is ir.LoopOrder.PARALLEL:
                add(per_statement(stmts), comp.order, (ci, bi), None)
                continue
            if cross_column(stmts):
                # A sequential block that reads, at a horizontal offset, what it writes: the columns are not
                # independent, so the sweep cannot live inside one thread.
                # (1) Temporaries that do not take part in the sweep at all -- computed from fields the block does
                # not write, read at their own level only -- are the usual reason (a horizontal stencil of the
                # inputs feeding a vertical recurrence).  Their statements are taken out of the loop and run as a
                # PARALLEL block over the same interval first; every level holds the value the loop would have
                # put there.
                hoisted, rest = _loop_invariant_statements(stmts, {t.name for t in stencil.temporaries})
                if hoisted:
                    add(per_statement(hoisted), ir.LoopOrder.PARALLEL, (ci, bi, "invariant"), None)
                    stmts = rest
            if cross_column(stmts):
                # (2) What remains is executed the way the reference's numpy backend executes every sequential block:
                # level by level, statement by statement over the plane (npir_codegen.py:243-248) -- stages cut at
                # the cross-column dependencies as in a PARALLEL block, launched once per K level by the host.
                cur = None
                add(per_statement(stmts), comp.order, (ci, bi), (len(stages), comp.order, block.interval))
                cur = None  # nothing after the block joins the per-level loop
            else:
                add([stmts], comp.order, (ci, bi), None)

    # mapping, statement splitting, extents
    for stage in stages:
        column = False
        for nest in stage.nests:
            if stage.plane is not None:
                break  # one level per launch: every other level is complete in memory, threads are points
            nest_writes = {s.target.name for s in nest.stmts}
            if nest.order is not ir.LoopOrder.PARALLEL:
                column = True
            for s in nest.stmts:
                for e in _stmt_field_reads(s):
                    if (e.offset[2] != 0 or e.koffset is not None) and e.name in stage.written:
                        column = True
                        if nest.order is ir.LoopOrder.PARALLEL and e.name in nest_writes:
                            nest.split_statements = True
                            if any(x.loops for x in nest.stmts):
                                raise UnsupportedStencil("a PARALLEL block with a `while` loop and a vertical dependency "
                                                         "on its own writes")
        stage.mapping = "column" if column else "ijk"
        ext = None
        for nest in stage.nests:
            for s in nest.stmts:
                ext = s.extent if ext is None else analysis._union(ext, s.extent)
        stage.extent = ext or analysis.ZERO_EXTENT

    # storage class of the remaining temporaries
    temp_names = {t.name for t in stencil.temporaries}
    where: Dict[str, Set[int]] = {}  # name -> ids of nests touching it
    nest_list = [(si, n) for si, st in enumerate(stages) for n in st.nests]
    bad_local: Set[str] = set()
    for nid, (si, nest) in enumerate(nest_list):
        defined: Set[str] = set()
        partial: Dict[str, ir.Expr] = {}  # assigned under a condition only so far: name -> that condition
        for s in nest.stmts:
            for e in _stmt_field_reads(s):
                if e.name in temp_names:
                    where.setdefault(e.name, set()).add(nid)
                    if e.offset != (0, 0, 0) or e.koffset is not None or e.name not in defined or nest.split_statements:
                        bad_local.add(e.name)
            name = s.target.name
            # written between the two branches and read by the condition: `not c` no longer complements the `c` the first
            # branch saw, the pair is not a full assignment
            for other in [n for n, c in partial.items() if n != name
                          and any(isinstance(e, ir.FieldAccess) and e.name == name for e in ir.walk(c))]:
                bad_local.add(other)
                del partial[other]
            if name in temp_names:
                where.setdefault(name, set()).add(nid)
                if s.target.offset != (0, 0, 0) or s.target.koffset is not None:
                    bad_local.add(name)  # written at another level than the one being computed
                if name in defined:
                    continue
                # A conditional assignment keeps the old value where the condition is false: unless the nest itself
                # assigned the name before, that is a value from an earlier nest or stage -- no thread-local register
                # holds it.  (Found by the fuzzer: `t = a` ended up one stage before `if c: t = b` and its reader.)
                # The two branches of one `if` / `else` together are a full assignment.
                if s.region is not None or s.loops:
                    bad_local.add(name)
                elif s.mask is None:
                    defined.add(name)
                    partial.pop(name, None)
                elif name in partial and s.mask == ir.UnaryOp("not", partial[name], s.mask.dtype):
                    defined.add(name)
                    del partial[name]
                elif name in partial:
                    bad_local.add(name)
                else:
                    partial[name] = s.mask
        bad_local |= set(partial)  # assigned under a condition only: what it held before shows through
    bad_local |= {t.name for t in stencil.temporaries if t.data_dims}  # several values per point: not a scalar
    bad_local |= {t.name for t in stencil.temporaries if tuple(t.axes) != ("I", "J", "K")}  # 2-d: outlives the K level
    local_names = {n for n in temp_names if n in where and n not in bad_local}
    # a local that is written in one nest and never read anywhere is dead but harmless
    scratch: Dict[str, Tuple[np.dtype, Extent2]] = {}
    temp_extents = analysis.storage_extents(stencil, extents)
    for t in stencil.temporaries:
        if t.name in where and t.name not in local_names:
            scratch[t.name] = (np.dtype(t.dtype), temp_extents[t.name])
    # (SSA values from the inliner are usually thread-local; one that is still needed after a stage cut --
    # e.g. by a run-time `if` further down -- is simply kept in scratch like any other temporary)

    # register forwarding in column stages: a value read exactly one level behind the sweep stays in a
    # register instead of being re-read from memory.  Exact only when (a) every vertical-offset read
    # of the name in the stage is that one pattern and (b) its writers and those readers cover the same
    # columns (otherwise a skipped write would leave a stale register).
    forwarded: Dict[Tuple[int, str], int] = {}
    for si, stage in enumerate(stages):
        if stage.mapping != "column":
            continue
        patterns: Dict[str, Set[Tuple[ir.LoopOrder, Tuple[int, int, int]]]] = {}
        extents_of: Dict[str, Set[Extent2]] = {}
        for nest in stage.nests:
            for s in nest.stmts:
                if s.target.name in stage.written:
                    extents_of.setdefault(s.target.name, set()).add(s.extent)
                for e in _stmt_field_reads(s):
                    if e.koffset is not None and e.name in stage.written:
                        patterns.setdefault(e.name, set()).add((nest.order, ("variable",)))
                    if e.offset[2] != 0 and e.name in stage.written:
                        patterns.setdefault(e.name, set()).add((nest.order, e.offset))
                        extents_of.setdefault(e.name, set()).add(s.extent)
        with_data_dims = {d.name for d in (*stencil.fields, *stencil.temporaries) if d.data_dims}
        unsafe = with_data_dims | {s.target.name for nest in stage.nests for s in nest.stmts
                  if s.mask is not None or s.region is not None or s.loops or s.target.offset != (0, 0, 0)
                  or s.target.koffset is not None}  # conditional / displaced writes
        for name, pats in patterns.items():
            if name in local_names or name in unsafe or len(extents_of.get(name, ())) != 1:
                continue
            if pats == {(ir.LoopOrder.FORWARD, (0, 0, -1))}:
                forwarded[(si, name)] = -1
            elif pats == {(ir.LoopOrder.BACKWARD, (0, 0, 1))}:
                forwarded[(si, name)] = 1

    # How each back-reading nest gets the register's first value, and which temporaries can live in
    # registers alone.  "carried": the previous nest of the chain left it there (statically adjacent
    # intervals, statically non-empty); "prime_if_prev_empty": adjacent, but the previous interval may be
    # empty at run time; "prime": load the level behind the first one from memory.
    def _static_length(iv: ir.Interval) -> Optional[int]:
        return iv.end.offset - iv.start.offset if iv.start.level is iv.end.level else None

    touched_in: Dict[str, Set[int]] = {}
    for si, stage in enumerate(stages):
        for nest in stage.nests:
            for st in nest.stmts:
                touched_in.setdefault(st.target.name, set()).add(si)
                for e in _stmt_field_reads(st):
                    touched_in.setdefault(e.name, set()).add(si)
    prime: Dict[Tuple[int, int, str], Tuple[str, Optional[int]]] = {}
    register_only: Set[str] = set()
    for (si, name), back in forwarded.items():
        stage = stages[si]
        want = ir.LoopOrder.FORWARD if back == -1 else ir.LoopOrder.BACKWARD
        needs_memory = name not in scratch or touched_in.get(name, set()) != {si}
        prev: Optional[int] = None
        for ni, nest in enumerate(stage.nests):
            writes = [idx for idx, st in enumerate(nest.stmts) if st.target.name == name]
            reads = [(idx, e) for idx, st in enumerate(nest.stmts) for e in _stmt_field_reads(st) if e.name == name]
            if not writes and not reads:
                continue
            if nest.order is not want:
                needs_memory = True
                if writes:
                    prev = None
                continue
            if any(e.offset == (0, 0, 0) and (not writes or idx <= writes[0]) for idx, e in reads):
                needs_memory = True  # reads the current level before this iteration wrote it
            if any(e.offset == (0, 0, back) for _, e in reads):
                mode = "prime"
                if prev is not None:
                    piv = stage.nests[prev].interval
                    adjacent = piv.end == nest.interval.start if back == -1 else piv.start == nest.interval.end
                    if adjacent:
                        n = _static_length(piv)
                        mode = "carried" if (n is not None and n > 0) else "prime_if_prev_empty"
                if mode != "carried":
                    needs_memory = True
                prime[(si, ni, name)] = (mode, prev)
                if not writes:
                    needs_memory = True  # the register is refilled from memory level by level
            prev = ni
        if not needs_memory:
            register_only.add(name)
    for name in register_only:
        del scratch[name]

    used: Set[str] = set()
    for _, nest in nest_list:
        for s in nest.stmts:
            used.add(s.target.name)
            for ex in ([c for _, c in s.loops] + ([s.value] if s.mask is None else [s.mask, s.value]) + _target_exprs(s.target)):
                for e in ir.walk(ex):
                    if isinstance(e, (ir.FieldAccess, ir.ScalarAccess)):
                        used.add(e.name)
    api_fields = [f for f in stencil.fields if f.name in used]
    params = [p for p in stencil.params if p.name in used]
    top_cache = _plan_top_cache(stencil, stages, scratch, forwarded, prime, local_names, register_only, touched_in)
    return Plan(stencil, stages, {**extents.fields, **temp_extents}, local_names, scratch, forwarded, prime, register_only,
                api_fields, params, top_cache, shared_forms)


def _end_relative(iv: ir.Interval) -> Optional[Tuple[Optional[int], int]]:
    """The part of ``iv`` that can overlap the levels counted from the END of the column: (first, one past the last) as
    offsets <= 0 from END, first = None when the interval starts at a START-relative level (below all of them); None when
    the interval ends at a START-relative level."""
    if iv.end.level is not ir.Level.END:
        return None
    return (iv.start.offset if iv.start.level is ir.Level.END else None), iv.end.offset


def _levels_covered(reads, writes) -> bool:
    """True when every END-relative level of ``reads`` lies in one of ``writes`` (ranges of ``_end_relative``)."""
    finite = [b for r in (*reads, *writes) for b in r if b is not None]
    if not finite:
        return True
    floor = min(finite) - 1  # stands for every level below the lowest END-relative bound

    def levels(rng):
        lo, hi = rng
        return range(floor if lo is None else lo, hi)

    covered = {k for w in writes for k in levels(w)}
    return all(k in covered for r in reads for k in levels(r))


def _plan_top_cache(stencil: ir.Stencil, stages: List[Stage], scratch, forwarded, prime, local_names, register_only,
                    touched_in) -> Dict[int, TopCache]:
    """Which fields of which column stages qualify for ``TopCache`` (see there)."""
    END0 = ir.AxisBound(ir.Level.END, 0)
    decls = {d.name: d for d in (*stencil.fields, *stencil.temporaries)}
    out: Dict[int, TopCache] = {}
    for si, stage in enumerate(stages):
        if stage.mapping != "column" or stage.plane is not None or len(stage.nests) < 2:
            continue
        if any(n.split_statements or n.order is ir.LoopOrder.PARALLEL for n in stage.nests):
            continue
        n_first = 0
        while n_first < len(stage.nests) and stage.nests[n_first].order is ir.LoopOrder.FORWARD:
            n_first += 1
        if n_first == 0 or n_first == len(stage.nests):
            continue
        if any(n.order is not ir.LoopOrder.BACKWARD for n in stage.nests[n_first:]):
            continue
        first, second = stage.nests[:n_first], stage.nests[n_first:]
        # the forward sweep ends at the last level, the backward sweep starts there; intervals adjacent within a sweep
        if first[-1].interval.end != END0 or second[0].interval.end != END0:
            continue
        if any(a.interval.end != b.interval.start for a, b in zip(first, first[1:])):
            continue
        if any(a.interval.start != b.interval.end for a, b in zip(second, second[1:])):
            continue
        written_first = {s.target.name for n in first for s in n.stmts}
        written_second = {s.target.name for n in second for s in n.stmts}
        names: List[str] = []
        for name in sorted(written_first - written_second):
            d = decls.get(name)
            if (d is None or name in local_names or name in register_only or tuple(d.axes) != ("I", "J", "K") or d.data_dims
                    or np.dtype(d.dtype).itemsize not in (4, 8)):
                continue
            ok = True
            read_back = False
            # Every cached level must be WRITTEN by the first sweep: the second sweep reads the cache, not memory.  The
            # cached levels lie above every START-relative interval bound (`margin` below), so exactly the first-sweep
            # nests whose interval has an END-relative bound can cover them: every END-relative level the second sweep
            # reads must lie in such a nest that assigns the field (the loop below rejects conditional / displaced
            # assignments).  A field written at the first level only and read back on all of them keeps its caller's
            # values above that level: not cacheable.
            written_levels = [_end_relative(n.interval) for n in first if any(st.target.name == name for st in n.stmts)]
            read_levels = [_end_relative(n.interval) for n in second
                           if any(e.name == name for st in n.stmts for e in _stmt_field_reads(st))]
            if not _levels_covered([r for r in read_levels if r], [w for w in written_levels if w]):
                ok = False
            for ni, nest in enumerate(stage.nests):
                for st in nest.stmts:
                    if st.target.name == name and (st.target.offset != (0, 0, 0) or st.target.koffset is not None
                                                   or st.mask is not None or st.region is not None or st.loops
                                                   or st.extent != stage.extent):
                        ok = False
                    for e in _stmt_field_reads(st):
                        if e.name != name:
                            continue
                        if e.koffset is not None or e.offset[:2] != (0, 0) or st.extent != stage.extent or st.region is not None or st.loops:
                            ok = False
                        elif ni >= n_first:
                            read_back = True
                            if e.offset != (0, 0, 0):
                                ok = False
                        elif e.offset == (0, 0, -1):
                            # the level behind the sweep: only through the forwarded register, never from memory
                            if forwarded.get((si, name)) != -1 or prime.get((si, ni, name), ("carried", None))[0] == "prime":
                                ok = False
                        elif e.offset != (0, 0, 0):
                            ok = False
            if ok and read_back:
                names.append(name)
        if not names or len(names) > 4:
            continue
        margin = 0
        for n in stage.nests:
            for b in (n.interval.start, n.interval.end):
                if b.level is ir.Level.START:
                    margin = max(margin, b.offset)
        store_through = {n for n in names if n not in scratch or touched_in.get(n, set()) != {si}}
        out[si] = TopCache(tuple(names), store_through, n_first, margin)
    return out
