"""A small typed stencil IR: what the minimal GTScript recogniser produces and what the ``hip:mi300``
backend pattern-matches against its hand-written kernel families.

This is deliberately NOT the reference's GTIR/OIR tower (SURVEY.md section 2 rows 9-11 are out of scope).
It keeps exactly the information that decides *values* and *shapes* on the hot path:

* computations in program order, each with an iteration order and K intervals
  (gtscript_frontend.py:1099-1167 of the reference),
* assignments of expression trees exactly as Python parses them (no re-association),
* per-node dtypes after the reference's promotion rules (gtir_upcaster.py:43-143) with explicit
  ``Cast`` nodes.
"""

from __future__ import annotations

import enum
from dataclasses import dataclass, field, replace
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np


class LoopOrder(str, enum.Enum):
    PARALLEL = "parallel"
    FORWARD = "forward"
    BACKWARD = "backward"


class Level(str, enum.Enum):
    START = "start"
    END = "end"


@dataclass(frozen=True)
class AxisBound:
    level: Level
    offset: int = 0

    def resolve(self, size: int) -> int:
        return self.offset if self.level is Level.START else size + self.offset


@dataclass(frozen=True)
class Interval:
    start: AxisBound
    end: AxisBound

    @classmethod
    def full(cls) -> "Interval":
        return cls(AxisBound(Level.START, 0), AxisBound(Level.END, 0))

    def range(self, size: int) -> Tuple[int, int]:
        return self.start.resolve(size), self.end.resolve(size)


# ---- expressions -------------------------------------------------------------------------------
@dataclass(frozen=True)
class Expr:
    pass


@dataclass(frozen=True)
class Literal(Expr):
    value: Union[bool, int, float]
    dtype: Optional[np.dtype]


@dataclass(frozen=True)
class FieldAccess(Expr):
    """``name[i, j, k]``.  ``koffset`` is a run-time integer expression ADDED to the K index
    (``field[0, 0, index]``, the reference's VariableKOffset: gtc/common.py; numpy: ``lk + k``,
    gtc/numpy/npir_codegen.py:110, 271-278); ``offset[2]`` is 0 then.  With ``absolute_k`` the expression
    IS the K index, counted from the field's K origin (``field.at(K=index)``, the reference's AbsoluteKIndex:
    gtc/common.py:356-380, debug_codegen.py:300-311); I and J are centred and the access is read-only."""

    name: str
    offset: Tuple[int, int, int]
    dtype: Optional[np.dtype] = None
    koffset: Optional[Expr] = None
    #: constant index into the field's data dimensions (``field[0, 0, 0][1, 0]``); () for plain fields.  Only
    #: inside the parser None stands for "not written": such statements are unrolled over every index
    #: (vector-valued assignments, /root/reference/tests/.../test_suites.py:980-1060).
    #: An element may also be an integer EXPRESSION evaluated at run time (``field[0, 0, 0][index]``).
    data_index: Optional[Tuple[Union[int, Expr], ...]] = ()
    absolute_k: bool = False


@dataclass(frozen=True)
class AxisIndex(Expr):
    """The value of the K iteration index, counted from the start of the compute domain (``K`` used as a
    value: the reference's IteratorAccess, gtscript_frontend.py:1312-1316; debug_codegen: the loop variable)."""

    axis: str
    dtype: Optional[np.dtype] = None


@dataclass(frozen=True)
class ScalarAccess(Expr):
    name: str
    dtype: Optional[np.dtype] = None


@dataclass(frozen=True)
class UnaryOp(Expr):
    op: str  # "-", "+", "not"
    expr: Expr
    dtype: Optional[np.dtype] = None


@dataclass(frozen=True)
class BinaryOp(Expr):
    op: str  # + - * / % ** > < >= <= == != and or
    left: Expr
    right: Expr
    dtype: Optional[np.dtype] = None


@dataclass(frozen=True)
class TernaryOp(Expr):
    cond: Expr
    true_expr: Expr
    false_expr: Expr
    dtype: Optional[np.dtype] = None


@dataclass(frozen=True)
class Cast(Expr):
    expr: Expr
    dtype: np.dtype


@dataclass(frozen=True)
class NativeCall(Expr):
    func: str
    args: Tuple[Expr, ...]
    dtype: Optional[np.dtype] = None


ARITHMETIC_OPS = ("+", "-", "*", "/", "%", "**")
COMPARISON_OPS = (">", "<", ">=", "<=", "==", "!=")
LOGICAL_OPS = ("and", "or")


# ---- statements --------------------------------------------------------------------------------
@dataclass(frozen=True)
class HorizontalInterval:
    """Half-open index range on one horizontal axis relative to the compute domain; None = unbounded
    (gtc/common.py ``HorizontalInterval``).  ``I[0]`` is START+0, ``I[-1]`` is END-1."""

    start: Optional[AxisBound] = None
    end: Optional[AxisBound] = None

    def overlap(self, extent: Tuple[int, int]) -> Optional[Tuple[int, int]]:
        """Distances from the block extent to the domain edges where the interval overlaps it, else None.
        Restates ``_overlap_along_axis`` (gtc/passes/horizontal_masks.py:15-47)."""
        start_diff: Optional[int]
        end_diff: Optional[int]
        if self.start is None:
            start_diff = 1000
        elif self.start.level is Level.START:
            start_diff = extent[0] - self.start.offset
        else:
            start_diff = None
        if self.end is None:
            end_diff = -1000
        elif self.end.level is Level.END:
            end_diff = extent[1] - self.end.offset
        else:
            end_diff = None
        if start_diff is not None and start_diff > 0 and end_diff is None and self.end is not None:
            if self.end.offset <= extent[0]:
                return None
        elif end_diff is not None and end_diff < 0 and start_diff is None and self.start is not None:
            if self.start.offset > extent[1]:
                return None
        return (min(start_diff, 0) if start_diff is not None else -10000,
                max(end_diff, 0) if end_diff is not None else 10000)

    def clip(self, lo: int, hi: int, size: int) -> Tuple[int, int]:
        """[lo, hi) (domain-relative, hi already includes ``size``) intersected with the interval."""
        if self.start is not None:
            lo = max(lo, self.start.resolve(size))
        if self.end is not None:
            hi = min(hi, self.end.resolve(size))
        return lo, hi


@dataclass(frozen=True)
class Region:
    """``with horizontal(region[i, j])``: the statement only runs where the mask holds."""

    i: HorizontalInterval
    j: HorizontalInterval


@dataclass(frozen=True)
class Assign:
    """``target = value``, or ``target = where(mask, value, target)`` when ``mask`` is set.

    Run-time ``if`` statements are flattened into masked assignments the way the reference lowers them
    (gtc/gtir_to_oir.py:146-218: a boolean mask temporary assigned once, then ``MaskStmt`` bodies;
    gtc/numpy/npir_codegen.py:205-210: ``np.where(mask, right, left)``).  ``group`` ties together the
    statements that came from ONE top-level statement of the interval block: the reference puts them
    in one HorizontalExecution (gtir_to_oir.py:225-232), so they share one compute extent.
    """

    target: FieldAccess
    value: Expr
    mask: Optional[Expr] = None
    group: int = -1
    region: Optional[Region] = None
    #: enclosing ``while`` loops, outermost first: (loop id, condition).  The condition already carries the
    #: enclosing masks, as the reference builds it (gtc/numpy/oir_to_npir.py:176-185: cond = mask AND cond;
    #: npir_codegen.py:252-267: ``while np.any(cond): body`` with every body statement masked by cond), and so
    #: does ``mask``.  Consecutive statements sharing a loop id are that loop's body.
    #: Loop ids >= ``POINTWISE_LOOP`` (backend option ``while_loops="pointwise"``): the loop is the compiled backends' per-point
    #: ``while (cond) { body }`` (gtc/gtcpp/gtcpp_codegen.py:257, gtc/debug/debug_codegen.py:138-144) -- the body statements'
    #: ``mask`` does NOT contain the condition, and an executor that works statement-wise over arrays has to apply the
    #: condition as it stood when the iteration began.
    loops: Tuple[Tuple[int, Expr], ...] = ()


POINTWISE_LOOP = 1 << 20


def stmt_exprs(stmt: "Assign"):
    """The expressions a statement reads: loop conditions and mask first (evaluated first), then the value."""
    for _, cond in stmt.loops:
        yield cond
    if stmt.mask is not None:
        yield stmt.mask
    yield stmt.value
    # run-time K offset / data index of the write itself
    if stmt.target.koffset is not None:
        yield stmt.target.koffset
    for d in stmt.target.data_index or ():
        if isinstance(d, Expr):
            yield d


def stmt_reads(stmt: "Assign"):
    """Every expression node read by ``stmt`` (pre-order over mask and value)."""
    for e in stmt_exprs(stmt):
        yield from walk(e)


@dataclass(frozen=True)
class IntervalBlock:
    interval: Interval
    body: Tuple[Assign, ...]


@dataclass(frozen=True)
class Computation:
    order: LoopOrder
    blocks: Tuple[IntervalBlock, ...]


@dataclass(frozen=True)
class FieldDecl:
    name: str
    dtype: Optional[np.dtype]  # None = AUTO (temporary not yet typed)
    axes: Tuple[str, ...] = ("I", "J", "K")
    data_dims: Tuple[int, ...] = ()
    is_api: bool = True


@dataclass(frozen=True)
class ScalarDecl:
    name: str
    dtype: np.dtype


@dataclass(frozen=True)
class Stencil:
    name: str
    fields: Tuple[FieldDecl, ...]  # API fields in signature order
    params: Tuple[ScalarDecl, ...]  # scalar parameters in signature order
    temporaries: Tuple[FieldDecl, ...]
    computations: Tuple[Computation, ...]

    def decl(self, name: str) -> Union[FieldDecl, ScalarDecl]:
        for d in (*self.fields, *self.temporaries, *self.params):
            if d.name == name:
                return d
        raise KeyError(name)

    def statements(self):
        """(computation, block, assign) triples in program order."""
        for comp in self.computations:
            for block in comp.blocks:
                for stmt in block.body:
                    yield comp, block, stmt


def walk(expr: Expr):
    """Pre-order traversal of an expression tree."""
    yield expr
    if isinstance(expr, FieldAccess):
        if expr.koffset is not None:
            yield from walk(expr.koffset)
        for d in expr.data_index or ():
            if isinstance(d, Expr):
                yield from walk(d)
    if isinstance(expr, (UnaryOp, Cast)):
        yield from walk(expr.expr)
    elif isinstance(expr, BinaryOp):
        yield from walk(expr.left)
        yield from walk(expr.right)
    elif isinstance(expr, TernaryOp):
        yield from walk(expr.cond)
        yield from walk(expr.true_expr)
        yield from walk(expr.false_expr)
    elif isinstance(expr, NativeCall):
        for a in expr.args:
            yield from walk(a)


def map_expr(expr: Expr, fn):
    """Rebuild ``expr`` bottom-up, applying ``fn`` to every rebuilt node."""
    if isinstance(expr, FieldAccess):
        if expr.koffset is not None:
            expr = replace(expr, koffset=map_expr(expr.koffset, fn))
        if any(isinstance(d, Expr) for d in expr.data_index or ()):
            expr = replace(expr, data_index=tuple(map_expr(d, fn) if isinstance(d, Expr) else d for d in expr.data_index))
    elif isinstance(expr, (UnaryOp, Cast)):
        expr = replace(expr, expr=map_expr(expr.expr, fn))
    elif isinstance(expr, BinaryOp):
        expr = replace(expr, left=map_expr(expr.left, fn), right=map_expr(expr.right, fn))
    elif isinstance(expr, TernaryOp):
        expr = replace(expr, cond=map_expr(expr.cond, fn), true_expr=map_expr(expr.true_expr, fn),
                       false_expr=map_expr(expr.false_expr, fn))
    elif isinstance(expr, NativeCall):
        expr = replace(expr, args=tuple(map_expr(a, fn) for a in expr.args))
    return fn(expr)


def fmt(expr: Expr) -> str:
    """Readable one-line rendering (used in error messages and ``StencilObject.source``)."""
    if isinstance(expr, Literal):
        return f"{expr.dtype}({expr.value!r})" if expr.dtype is not None else repr(expr.value)
    if isinstance(expr, FieldAccess):
        k = f"{expr.offset[2]}" if expr.koffset is None else ("K=" if expr.absolute_k else "") + fmt(expr.koffset)
        data = "".join(f"[{fmt(d) if isinstance(d, Expr) else d}]" for d in (expr.data_index or ()))
        return f"{expr.name}[{expr.offset[0]},{expr.offset[1]},{k}]{data}"
    if isinstance(expr, AxisIndex):
        return expr.axis
    if isinstance(expr, ScalarAccess):
        return expr.name
    if isinstance(expr, UnaryOp):
        return f"({expr.op}{fmt(expr.expr)})"
    if isinstance(expr, BinaryOp):
        return f"({fmt(expr.left)} {expr.op} {fmt(expr.right)})"
    if isinstance(expr, TernaryOp):
        return f"({fmt(expr.true_expr)} if {fmt(expr.cond)} else {fmt(expr.false_expr)})"
    if isinstance(expr, Cast):
        return f"{expr.dtype}({fmt(expr.expr)})"
    if isinstance(expr, NativeCall):
        return f"{expr.func}({', '.join(fmt(a) for a in expr.args)})"
    return repr(expr)
