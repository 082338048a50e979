"""ORACLE (test infrastructure, NOT product code): an INDEPENDENT interpreter of GTScript definitions.

Why: ``oracle/numpy_backend.py`` interprets the PRODUCT's parsed IR, so a parse / extent / dtype-rule error in
``gt4py_amd.cartesian.frontend`` or ``analysis`` would be common to the product and its oracle.  This module
shares NOTHING with the product: it reads the Python source of a definition function with ``ast`` and executes it on numpy arrays by
the rules of the reference, each restated from the reference's own source:

* one HORIZONTAL EXECUTION per top-level statement of an interval block -- an ``if`` / ``while`` / ``with horizontal`` block with
  everything inside is ONE (/root/reference/src/gt4py/cartesian/gtc/gtir_to_oir.py:224-232; field-``if``: a mask temporary assigned
  first, then masked statements, :146-188);
* EXTENTS: statements visited last to first; the block extent of an execution is the union of the extents so far required of the
  fields it writes; every read then requires ``block + offset`` of its field
  (gtc/passes/oir_optimizations/utils.py:250-313); reads under a horizontal region require only what the region's position
  relative to the domain edges makes reachable (:52-76, gtc/passes/horizontal_masks.py:15-58);
* EXECUTION (gtc/numpy/npir_codegen.py:281-330, gtc/numpy/oir_to_npir.py:147-215): interval blocks in program order; PARALLEL: each
  statement over the block's whole (extended) box and K range before the next; FORWARD / BACKWARD: level by level, all statements
  of the block per level; masked assignment ``where(mask, rhs, left)``; a region restricts the box to its intersection with the
  extended domain;
* DTYPES (gtc/passes/gtir_upcaster.py:41-143, gtc/common.py:105-127,287-306,513-533,561-569,600-640): literals are int64 / float64
  (``literal_int`` / ``literal_float`` bits); an operator's operands are cast to the lowest-ranking loop of ITS numpy ufunc that
  every operand fits (so int64 + float32 -> float32, not numpy's float64); ternary branches to the higher-ranking of the two; ``**``
  and the cast functions leave their operands alone; an assignment casts to the target; a temporary takes the dtype of its first
  assignment (gtc/passes/gtir_dtype_resolver.py:16-71).

Two places where the reference's BACKENDS disagree with each other, and what is done here:

* ``while``: the numpy backend re-evaluates the loop condition as the mask of EVERY body statement (oir_to_npir.py:187-196), so a
  statement after one that falsifies the condition is skipped for that point; the gt:* backends emit an ordinary per-point loop
  (OIR's ``While`` carries no mask of its own, gtir_to_oir.py:136-144).  ``while_semantics="numpy"`` (the default: the numpy backend is
  the oracle SURVEY.md section 8c names, and ``hip:mi300`` follows it) or ``"pointwise"``; ``while_reevaluates`` is non-zero when the
  two differ for the program at hand, so that a test can say its result is backend-defined there.
* extents of a LATER interval block that reads, at a horizontal offset, a temporary an EARLIER block of the same (merged) vertical
  loop wrote: the reference visits merged blocks in forward order (utils.py:266-268 over the sections AdjacentLoopMerging glued,
  oir_optimizations/vertical_loop_merging.py:17-38), so the earlier block's writes keep a zero extent and the cells next to the
  domain stay uninitialised (``Field.empty``).  This interpreter visits everything last to first (every needed cell is computed);
  ``forward_section_quirk`` lists the temporaries for which the reference's order would leave cells undefined.

Only ``tests/`` may import this module.  It never imports ``gt4py_amd``.
"""

from __future__ import annotations

import ast
import inspect
import textwrap
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import scipy.special


class Unsupported(Exception):
    """The definition uses a construct this interpreter does not restate (the test skips it and counts it)."""


# gtc/common.py:105-118 (DataType ids give the ranking)
_RANK = {np.dtype(np.bool_): 10, np.dtype(np.int8): 11, np.dtype(np.int16): 12, np.dtype(np.int32): 14, np.dtype(np.int64): 18,
         np.dtype(np.float32): 104, np.dtype(np.float64): 108}
_BIN_UFUNC = {ast.Add: np.add, ast.Sub: np.subtract, ast.Mult: np.multiply, ast.Div: np.true_divide}
_CMP_UFUNC = {ast.Gt: np.greater, ast.Lt: np.less, ast.GtE: np.greater_equal, ast.LtE: np.less_equal, ast.Eq: np.equal, ast.NotEq: np.not_equal}
_NATIVE = {  # gtc/common.py:954-990 + gtc/ufuncs.py
    "abs": np.abs, "min": np.minimum, "max": np.maximum, "mod": np.remainder, "sin": np.sin, "cos": np.cos, "tan": np.tan,
    "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan, "arcsin": np.arcsin, "arccos": np.arccos, "arctan": np.arctan,
    "sinh": np.sinh, "cosh": np.cosh, "tanh": np.tanh, "asinh": np.arcsinh, "acosh": np.arccosh, "atanh": np.arctanh,
    "arcsinh": np.arcsinh, "arccosh": np.arccosh, "arctanh": np.arctanh, "sqrt": np.sqrt, "exp": np.exp, "log": np.log,
    "log10": np.log10, "cbrt": np.cbrt, "isfinite": np.isfinite, "isinf": np.isinf, "isnan": np.isnan, "floor": np.floor,
    "ceil": np.ceil, "trunc": np.trunc, "round": np.round, "round_away_from_zero": lambda x: np.copysign(np.floor(np.abs(x) + 0.5), x),
    "erf": scipy.special.erf, "erfc": scipy.special.erfc, "gamma": scipy.special.gamma,
}
_CASTS = {"int32": np.int32, "int64": np.int64, "float32": np.float32, "float64": np.float64}
_FLOAT_ONLY_TYPES = ["f->f", "d->d"]  # gtc/ufuncs.py gives erf / erfc / gamma / round these loops when they are not ufuncs


def _ufunc_targets(ufunc, dtypes) -> List[np.dtype]:
    """gtir_upcaster._numpy_ufunc_upcasting_rule: among the ufunc's loops whose every input type is a supported dtype that the
    operand's dtype does not outrank, the one with the lowest sum of ranks."""
    best: Dict[int, List[np.dtype]] = {}
    types = getattr(ufunc, "types", None) or _FLOAT_ONLY_TYPES
    if not isinstance(ufunc, np.ufunc) or ufunc in (scipy.special.erf, scipy.special.erfc, scipy.special.gamma):
        types = _FLOAT_ONLY_TYPES  # (gtc/ufuncs.py:16-40 gives these the loops f->f and d->d whatever scipy offers)
    for t in types:
        ins, out = t.split("->")
        if len(ins) != len(dtypes):
            continue
        cands = [np.dtype(c) for c in ins]
        if any(c not in _RANK for c in cands) or np.dtype(out[0]) not in _RANK:
            continue
        if all(_RANK[d] <= _RANK[c] for d, c in zip(dtypes, cands)):
            best[sum(_RANK[c] for c in cands)] = cands
    if not best:
        raise Unsupported(f"no loop of {ufunc} takes {dtypes}")
    return best[min(best)]


def _cast(value, dtype: np.dtype):
    if isinstance(value, np.ndarray):
        return value if value.dtype == dtype else value.astype(dtype)
    return value if np.asarray(value).dtype == dtype and isinstance(value, np.generic) else dtype.type(value)


def _dtype_of(value) -> np.dtype:
    return np.asarray(value).dtype


# ---- the program ----------------------------------------------------------------------------------------------------------------
@dataclass
class _Bound:
    end: bool  # measured from the END of the axis (else from its START)
    offset: int


@dataclass
class _Block:
    """One interval block of one computation."""
    order: str
    start: _Bound
    stop: _Bound
    stmts: List[ast.stmt]
    loop: int  # index of the `with computation` it belongs to


@dataclass
class _Field:
    name: str
    axes: Tuple[str, ...]
    api: bool
    dtype: Optional[np.dtype] = None
    data_dims: Tuple[int, ...] = ()
    frame: Optional[np.ndarray] = None  # (X, Y, Z, *data) with size-1 axes where the field has none


Extent = Tuple[Tuple[int, int], Tuple[int, int]]
_ZERO: Extent = ((0, 0), (0, 0))


def _union(a: Extent, b: Extent) -> Extent:
    return tuple((min(x[0], y[0]), max(x[1], y[1])) for x, y in zip(a, b))  # type: ignore[return-value]


_AXES_NAMES = {"IJK": ("I", "J", "K"), "IJ": ("I", "J"), "IK": ("I", "K"), "JK": ("J", "K"), "I": ("I",), "J": ("J",), "K": ("K",)}


class Interpreter:
    def __init__(self, definition, *, externals: Optional[Dict[str, Any]] = None, literal_int: int = 64, literal_float: int = 64,
                 while_semantics: str = "numpy"):
        assert while_semantics in ("numpy", "pointwise")
        self.while_semantics = while_semantics
        self._while_differs_counted = False
        self.fn = definition
        src = textwrap.dedent(inspect.getsource(definition))
        tree = ast.parse(src)
        self.fdef = next(n for n in tree.body if isinstance(n, ast.FunctionDef))
        self.globals = dict(getattr(definition, "__globals__", {}))
        try:  # a definition nested in another function sees that function's locals
            self.globals.update(inspect.getclosurevars(definition).nonlocals)
        except (TypeError, ValueError):
            pass
        self.externals = dict(externals or {})
        self.int_t = np.dtype(np.int64 if literal_int == 64 else np.int32)
        self.float_t = np.dtype(np.float64 if literal_float == 64 else np.float32)
        self.params: List[Tuple[str, Optional[Tuple[str, ...]], Any]] = []  # (name, axes or None for a scalar, annotation ast)
        args = self.fdef.args
        for a in [*args.posonlyargs, *args.args, *args.kwonlyargs]:
            self.params.append((a.arg, self._annotation_axes(a.annotation), a.annotation))
        self._typed: Dict[str, Tuple[np.dtype, Tuple[str, ...], Tuple[int, ...]]] = {}
        self.blocks = self._blocks()
        self.while_reevaluates = 0
        self.forward_section_quirk: List[str] = []

    # ---- signature -----------------------------------------------------------------------------------------------------------
    def _annotation_axes(self, ann) -> Optional[Tuple[str, ...]]:
        """Axes of a ``Field[...]`` parameter, None for a scalar.  Syntactic where the annotation is spelled out; an alias
        (``Field3D``) is looked up in the definition's globals and only its ``axes`` attribute read."""
        if isinstance(ann, ast.Constant) and isinstance(ann.value, str):  # a string annotation
            return self._annotation_axes(ast.parse(ann.value, mode="eval").body)
        if isinstance(ann, ast.Name) and isinstance(self.globals.get(ann.id), str):  # ... or an alias of one
            return self._annotation_axes(ast.parse(self.globals[ann.id], mode="eval").body)
        if isinstance(ann, ast.Subscript) and isinstance(ann.value, ast.Name) and ann.value.id == "GlobalTable":
            return ()  # data dimensions only
        if isinstance(ann, ast.Subscript) and isinstance(ann.value, ast.Name) and ann.value.id == "Field":
            first = ann.slice.elts[0] if isinstance(ann.slice, ast.Tuple) else ann.slice
            if isinstance(first, ast.Name) and first.id in _AXES_NAMES:
                return _AXES_NAMES[first.id]
            return ("I", "J", "K")
        if isinstance(ann, ast.Name) and ann.id in self.globals and hasattr(self.globals[ann.id], "axes"):
            return tuple(str(x) for x in self.globals[ann.id].axes)
        return None

    def _scalar_dtype(self, ann, value) -> np.dtype:
        txt = ast.unparse(ann) if ann is not None else ""
        for key, dt in (("float32", np.float32), ("float64", np.float64), ("int32", np.int32), ("int64", np.int64), ("int8", np.int8),
                        ("int16", np.int16), ("bool", np.bool_)):
            if key in txt:
                return np.dtype(dt)
        if txt == "float":
            return np.dtype(np.float64)
        if txt == "int":
            return np.dtype(np.int64)
        if isinstance(value, (bool, np.bool_)):
            return np.dtype(np.bool_)
        if isinstance(value, (int, np.integer)):
            return np.dtype(np.int64)
        return np.dtype(np.float64)

    # ---- structure -----------------------------------------------------------------------------------------------------------
    def _blocks(self) -> List[_Block]:
        out: List[_Block] = []
        loop = 0
        for node in self.fdef.body:
            if isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant):
                continue  # docstring
            if isinstance(node, ast.ImportFrom):
                if not (node.module or "").endswith(("__externals__", "__gtscript__")):
                    raise Unsupported("import in a definition")
                continue
            if isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
                # a typed temporary (gtscript_frontend.py:2243-2263); with a value: one PARALLEL computation over the whole
                # column that assigns it comes first (:809-850)
                self._typed[node.target.id] = self._declared(node.annotation)
                if node.value is not None:
                    init = ast.Assign(targets=[ast.Name(id=node.target.id, ctx=ast.Store())], value=node.value)
                    out.insert(len([b for b in out if b.loop < 0]), _Block("PARALLEL", _Bound(False, 0), _Bound(True, 0),
                                                                           [ast.fix_missing_locations(ast.copy_location(init, node))], -1))
                continue
            if not isinstance(node, ast.With):
                raise Unsupported(f"top-level {type(node).__name__}")
            order, interval = None, None
            for item in node.items:
                call = item.context_expr
                if not isinstance(call, ast.Call) or not isinstance(call.func, ast.Name):
                    raise Unsupported("with item")
                if call.func.id == "computation":
                    arg = call.args[0] if call.args else call.keywords[0].value
                    order = arg.id
                elif call.func.id == "interval":
                    interval = self._interval(call)
                else:
                    raise Unsupported(f"with {call.func.id}")
            if order is None:
                raise Unsupported("with block without computation")
            if interval is not None:
                out.append(_Block(order, interval[0], interval[1], node.body, loop))
            else:
                for inner in node.body:
                    if not (isinstance(inner, ast.With) and len(inner.items) == 1 and isinstance(inner.items[0].context_expr, ast.Call)
                            and getattr(inner.items[0].context_expr.func, "id", None) == "interval"):
                        raise Unsupported("statement outside an interval")
                    iv = self._interval(inner.items[0].context_expr)
                    out.append(_Block(order, iv[0], iv[1], inner.body, loop))
            loop += 1
        for b in out:
            b.stmts = self._inline_compile_time_ifs(list(b.stmts))
        return out

    def _inline_compile_time_ifs(self, stmts):
        out = []
        for s in stmts:
            if isinstance(s, ast.If) and isinstance(s.test, ast.Call) and getattr(s.test.func, "id", None) == "__INLINED":
                names = {**{k: v for k, v in self.globals.items() if isinstance(v, (int, float, bool, np.generic))}, **self.externals}
                chosen = s.body if eval(compile(ast.Expression(body=s.test.args[0]), "<inlined>", "eval"), {"__builtins__": {}}, names) else s.orelse
                out.extend(self._inline_compile_time_ifs(chosen))
                continue
            for attr in ("body", "orelse"):
                if isinstance(s, (ast.If, ast.While, ast.With)) and getattr(s, attr, None):
                    setattr(s, attr, self._inline_compile_time_ifs(getattr(s, attr)))
            out.append(s)
        return out

    def _const_int(self, node) -> int:
        if isinstance(node, ast.Constant) and isinstance(node.value, int):
            return int(node.value)
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
            return -self._const_int(node.operand)
        if isinstance(node, ast.Name):
            v = self.externals.get(node.id, self.globals.get(node.id))
            if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
                return int(v)
        if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub)):
            a, b = self._const_int(node.left), self._const_int(node.right)
            return a + b if isinstance(node.op, ast.Add) else a - b
        raise Unsupported(f"not a compile-time integer: {ast.unparse(node)}")

    def _interval(self, call: ast.Call) -> Tuple[_Bound, _Bound]:
        args = list(call.args)
        if len(args) == 1 and isinstance(args[0], ast.Constant) and args[0].value is Ellipsis:
            return _Bound(False, 0), _Bound(True, 0)
        if len(args) != 2:
            raise Unsupported("interval form")

        def bound(node, is_stop):
            if isinstance(node, ast.Constant) and node.value is None:
                return _Bound(True, 0) if is_stop else _Bound(False, 0)
            v = self._const_int(node)
            return _Bound(True, v) if v < 0 else _Bound(False, v)

        return bound(args[0], False), bound(args[1], True)

    # ---- execution -----------------------------------------------------------------------------------------------------------
    def __call__(self, fields: Dict[str, Tuple[np.ndarray, Tuple[int, ...]]], scalars: Dict[str, Any], domain: Tuple[int, int, int]):
        """``fields[name] = (array, origin)``: arrays are modified in place, exactly the cells the reference would write."""
        self.domain = tuple(int(d) for d in domain)
        ni, nj, nk = self.domain
        self.flds: Dict[str, _Field] = {}
        self.scalars: Dict[str, Any] = {}
        for name, axes, ann in self.params:
            if axes is None:
                if name not in scalars:
                    raise Unsupported(f"scalar {name} not given")
                dt = self._scalar_dtype(ann, scalars[name])
                self.scalars[name] = dt.type(scalars[name])
            else:
                if fields.get(name) is None:
                    continue  # an optional field that was not passed: any access is an error below
                arr, origin = fields[name]
                f = _Field(name, axes, True, np.dtype(arr.dtype), tuple(arr.shape[len(axes):]))
                self.flds[name] = f
        # the frame: every array is embedded at a common padding H (horizontal) / HK (vertical), large enough for every offset
        # in the program applied on top of every extent, and for the halos the caller's arrays have
        offs = [abs(self._const_or_zero(e)) for n in ast.walk(self.fdef) if isinstance(n, ast.Subscript) for e in self._subscript_elts(n)]
        reach = (max(offs) if offs else 0) + 1
        n_stmt = sum(1 for b in self.blocks for _ in ast.walk(ast.Module(body=b.stmts, type_ignores=[])) if isinstance(_, (ast.Assign, ast.AugAssign, ast.AnnAssign)))
        self.H = min(reach * (n_stmt + 1), 40)
        self.HK = reach + 1
        fields = {k: v for k, v in fields.items() if v is not None}
        for name, (arr, origin) in fields.items():
            if name not in self.flds:
                continue
            f = self.flds[name]
            org = dict(zip(f.axes, origin))
            shp = dict(zip(f.axes, arr.shape))
            for ax, n in (("I", ni), ("J", nj)):
                if ax in f.axes:
                    self.H = max(self.H, org[ax], shp[ax] - org[ax] - n)
            if "K" in f.axes:
                self.HK = max(self.HK, org["K"], shp["K"] - org["K"] - nk)
        H, HK = self.H, self.HK
        self._api_slices: Dict[str, Tuple[slice, ...]] = {}
        for name, (arr, origin) in fields.items():
            if name not in self.flds:
                continue
            f = self.flds[name]
            f.frame = self._empty(f.axes, f.dtype, f.data_dims)
            sl = []
            org = dict(zip(f.axes, origin))
            shp = dict(zip(f.axes, arr.shape))
            for ax, pad in (("I", H), ("J", H), ("K", HK)):
                if ax in f.axes:
                    sl.append(slice(pad - org[ax], pad - org[ax] + shp[ax]))
                else:
                    sl.append(slice(0, 1))
            self._api_slices[name] = tuple(sl)
            view = arr.reshape(tuple(shp.get(ax, 1) for ax in "IJK") + f.data_dims)
            f.frame[tuple(sl)] = view
        self._hexecs = self._horizontal_executions()
        self._extents()
        for b_idx, block in enumerate(self.blocks):
            self._run_block(b_idx, block)
        for name, (arr, origin) in fields.items():
            if name in self.flds and name in self.written_api:
                f = self.flds[name]
                arr[...] = f.frame[self._api_slices[name]].reshape(arr.shape)

    def _empty(self, axes, dtype, data_dims=()):
        ni, nj, nk = self.domain
        shape = (ni + 2 * self.H if "I" in axes else 1, nj + 2 * self.H if "J" in axes else 1, nk + 2 * self.HK if "K" in axes else 1)
        a = np.empty(shape + tuple(data_dims), dtype=dtype)
        a[...] = np.nan if dtype.kind == "f" else (0 if dtype.kind != "b" else False)
        return a

    @staticmethod
    def _subscript_elts(n: ast.Subscript):
        s = n.slice
        return list(s.elts) if isinstance(s, ast.Tuple) else [s]

    def _const_or_zero(self, node) -> int:
        try:
            return self._const_int(node)
        except Unsupported:
            return 0

    # ---- horizontal executions and extents ------------------------------------------------------------------------------------
    # ---- statements over whole data dimensions: unrolled into one assignment per element (frontend/defir_to_gtir.py:140-290) --------------
    def _dims_of(self, name):
        if name in self.flds:
            return tuple(self.flds[name].data_dims)
        typed = self._typed.get(name)
        return tuple(typed[2]) if typed else ()

    def _vector_elements(self, e):
        """Nested lists of element expressions for an expression over vector / matrix fields, or the expression itself (a scalar)."""
        def whole(node):  # a vector field named without a data index -> (name, offset elts)
            if isinstance(node, ast.Name) and self._dims_of(node.id):
                return node.id, [ast.Constant(value=0)] * 3
            if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and self._dims_of(node.value.id):
                elts = self._elts3(node.value.id, self._subscript_elts(node))
                if elts is not None and len(elts) == 3:
                    return node.value.id, elts
            return None

        w = whole(e)
        if w is not None:
            name, off = w
            dims = self._dims_of(name)

            def ref(*idx):
                base = ast.Subscript(value=ast.Name(id=name, ctx=ast.Load()), slice=ast.Tuple(elts=list(off), ctx=ast.Load()), ctx=ast.Load())
                index = ast.Tuple(elts=[ast.Constant(value=i) for i in idx], ctx=ast.Load()) if len(idx) > 1 else ast.Constant(value=idx[0])
                return ast.Subscript(value=base, slice=index, ctx=ast.Load())

            if len(dims) == 1:
                return [ref(i) for i in range(dims[0])]
            if len(dims) == 2:
                return [[ref(r, c) for c in range(dims[1])] for r in range(dims[0])]
            raise Unsupported("statements over more than two data dimensions")
        if isinstance(e, ast.Attribute) and e.attr == "T":
            m = self._vector_elements(e.value)
            if isinstance(m, list) and m and isinstance(m[0], list):
                return [list(x) for x in zip(*m)]
        if isinstance(e, ast.BinOp):
            lhs, rhs = self._vector_elements(e.left), self._vector_elements(e.right)
            if isinstance(e.op, ast.MatMult):
                out = []
                for row in lhs:
                    acc = ast.BinOp(left=row[0], op=ast.Mult(), right=rhs[0])
                    for i in range(1, len(row)):
                        acc = ast.BinOp(left=acc, op=ast.Add(), right=ast.BinOp(left=row[i], op=ast.Mult(), right=rhs[i]))
                    out.append(acc)
                return out
            if isinstance(lhs, list) and isinstance(rhs, list):
                def zipped(a, b):
                    return [zipped(x, y) if isinstance(x, list) else ast.BinOp(left=x, op=e.op, right=y) for x, y in zip(a, b)]
                return zipped(lhs, rhs)
            if isinstance(lhs, list) or isinstance(rhs, list):
                def each(a, left_is_list):
                    return [each(x, left_is_list) if isinstance(x, list) else
                            (ast.BinOp(left=x, op=e.op, right=rhs) if left_is_list else ast.BinOp(left=lhs, op=e.op, right=x)) for x in a]
                return each(lhs, True) if isinstance(lhs, list) else each(rhs, False)
        if isinstance(e, ast.UnaryOp):
            inner = self._vector_elements(e.operand)
            if isinstance(inner, list):
                def neg(a):
                    return [neg(x) if isinstance(x, list) else ast.UnaryOp(op=e.op, operand=x) for x in a]
                return neg(inner)
        return e

    def _unroll_vector_statements(self, stmts):
        import itertools

        out = []
        for s in stmts:
            for attr in ("body", "orelse"):
                if isinstance(s, (ast.If, ast.While, ast.With)) and getattr(s, attr, None):
                    setattr(s, attr, self._unroll_vector_statements(getattr(s, attr)))
            if isinstance(s, ast.Assign) and len(s.targets) == 1:
                t = s.targets[0]
                name = t.id if isinstance(t, ast.Name) else (t.value.id if isinstance(t, ast.Subscript) and isinstance(t.value, ast.Name) else None)
                dims = self._dims_of(name) if name else ()
                is_whole = dims and (isinstance(t, ast.Name) or len(self._elts3(name, self._subscript_elts(t)) or ()) == 3)
                if is_whole:
                    off = [ast.Constant(value=0)] * 3 if isinstance(t, ast.Name) else self._elts3(name, self._subscript_elts(t))
                    value = self._vector_elements(s.value)
                    for index in itertools.product(*(range(d) for d in dims)):
                        v = value
                        if isinstance(v, list):
                            for i in index:
                                v = v[i]
                        base = ast.Subscript(value=ast.Name(id=name, ctx=ast.Load()), slice=ast.Tuple(elts=list(off), ctx=ast.Load()), ctx=ast.Store())
                        idx = ast.Tuple(elts=[ast.Constant(value=i) for i in index], ctx=ast.Load()) if len(index) > 1 else ast.Constant(value=index[0])
                        one = ast.Assign(targets=[ast.Subscript(value=base, slice=idx, ctx=ast.Store())], value=v)
                        out.append(ast.fix_missing_locations(ast.copy_location(one, s)))
                    continue
            out.append(s)
        return out

    def _horizontal_executions(self):
        """[(block index, statement)]: one per top-level statement of every interval block, in program order."""
        out = []
        for b, block in enumerate(self.blocks):
            block.stmts = self._unroll_vector_statements(list(block.stmts))
            for s in block.stmts:
                if isinstance(s, (ast.Pass, ast.ImportFrom)):
                    continue
                call = s.items[0].context_expr if isinstance(s, ast.With) and len(s.items) == 1 else None
                if isinstance(call, ast.Call) and getattr(call.func, "id", None) == "horizontal" and len(call.args) > 1:
                    # `with horizontal(r1, r2)`: the body once per region, one statement each (gtscript_frontend.py:1955-1977)
                    for arg in call.args:
                        one = ast.With(items=[ast.withitem(context_expr=ast.Call(func=call.func, args=[arg], keywords=[]))], body=s.body)
                        out.append((b, ast.fix_missing_locations(ast.copy_location(one, s))))
                else:
                    out.append((b, s))
        return out

    def _accesses(self, stmt, region=None, out=None):
        """(name, (di, dj), is_write, region) of every field access inside a statement, function calls expanded."""
        out = [] if out is None else out
        if isinstance(stmt, ast.With):
            reg = self._region(stmt)
            for s in stmt.body:
                self._accesses(s, reg, out)
            return out
        if isinstance(stmt, (ast.If, ast.While)):
            self._expr_accesses(stmt.test, region, out, {})
            for s in [*stmt.body, *stmt.orelse]:
                self._accesses(s, region, out)
            return out
        if isinstance(stmt, ast.Assign):
            if len(stmt.targets) != 1:
                raise Unsupported("chained assignment")
            targets, value = ([stmt.targets[0]], stmt.value)
            if isinstance(stmt.targets[0], ast.Tuple):
                targets = list(stmt.targets[0].elts)
        elif isinstance(stmt, ast.AugAssign):
            targets, value = [stmt.target], stmt.value
            self._expr_accesses(stmt.target, region, out, {})
        elif isinstance(stmt, ast.AnnAssign):
            targets, value = [stmt.target], stmt.value
        elif isinstance(stmt, (ast.Pass, ast.ImportFrom)):
            return out
        else:
            raise Unsupported(f"statement {type(stmt).__name__}")
        if value is not None:
            self._expr_accesses(value, region, out, {})
        for t in targets:
            name, off, _ = self._target(t)
            out.append((name, (0, 0), True, region))
            if not isinstance(off[2], int):
                self._expr_accesses(off[2], region, out, {})
        return out

    def _target(self, t):
        if isinstance(t, ast.Name):
            return t.id, (0, 0, 0), None
        if isinstance(t, ast.Subscript) and isinstance(t.value, ast.Name):
            elts = self._subscript_elts(t)
            elts = self._elts3(t.value.id, elts) or elts
            if len(elts) != 3 and self._is_field(t.value.id) and len(elts) == len(self._field_data_dims(t.value.id)):
                return t.value.id, (0, 0, 0), elts  # field[n] = ...: a data index at zero offset
            if len(elts) == 3:
                di, dj = self._const_int(elts[0]), self._const_int(elts[1])
                if di or dj:
                    raise Unsupported("write at a horizontal offset")
                try:
                    dk = self._const_int(elts[2])
                except Unsupported:
                    dk = elts[2]  # a run-time K offset: an expression
                return t.value.id, (0, 0, dk), None
        if isinstance(t, ast.Subscript) and isinstance(t.value, ast.Subscript):  # field[0, 0, 0][n]
            name, off, _ = self._target(t.value)
            return name, off, self._subscript_elts(t)
        raise Unsupported(f"assignment target {ast.unparse(t)}")

    def _expr_accesses(self, e, region, out, env, shift=(0, 0)):
        """Field reads of an expression; ``env`` maps the parameters / locals of a gtscript function being expanded to
        (expression, env, shift) of the caller."""
        if isinstance(e, ast.Name):
            if e.id in env:
                sub, sub_env, sub_shift = env[e.id]
                self._expr_accesses(sub, region, out, sub_env, (shift[0] + sub_shift[0], shift[1] + sub_shift[1]))
            elif self._is_field(e.id):
                out.append((e.id, shift, False, region))
            return
        if isinstance(e, ast.Subscript):
            base = e.value
            if isinstance(base, ast.Subscript):  # data index on top of an offset
                self._expr_accesses(base, region, out, env, shift)
                for x in self._subscript_elts(e):
                    self._expr_accesses(x, region, out, env, shift)
                return
            if isinstance(base, ast.Attribute) and base.attr == "A" and isinstance(base.value, ast.Name):
                out.append((base.value.id, shift, False, region))
                for x in self._subscript_elts(e):
                    self._expr_accesses(x, region, out, env, shift)
                return
            if isinstance(base, ast.Name):
                elts = self._subscript_elts(e)
                if base.id in env or self._is_field(base.id):
                    elts = self._elts3(base.id, elts) or elts
                    if len(elts) != 3:
                        if self._is_field(base.id) and len(elts) == len(self._field_data_dims(base.id)):
                            out.append((base.id, shift, False, region))  # data index at zero offset
                            return
                        raise Unsupported(f"offset form {ast.unparse(e)}")
                    di, dj = self._const_int(elts[0]), self._const_int(elts[1])
                    try:
                        self._const_int(elts[2])
                    except Unsupported:
                        self._expr_accesses(elts[2], region, out, env, shift)
                    s2 = (shift[0] + di, shift[1] + dj)
                    if base.id in env:
                        sub, sub_env, sub_shift = env[base.id]
                        self._expr_accesses(sub, region, out, sub_env, (s2[0] + sub_shift[0], s2[1] + sub_shift[1]))
                    else:
                        out.append((base.id, s2, False, region))
                    return
            raise Unsupported(f"subscript {ast.unparse(e)}")
        if isinstance(e, ast.Call):
            fn = self._gt_function(e)
            if fn is not None:
                for ret, fenv in self._expand_call(fn, e, env, shift):
                    self._expr_accesses(ret, region, out, fenv, (0, 0))
                return
            if isinstance(e.func, ast.Attribute) and isinstance(e.func.value, ast.Name):
                self._expr_accesses(e.func.value, region, out, env, shift)
            for a in [*e.args, *[kw.value for kw in e.keywords]]:
                self._expr_accesses(a, region, out, env, shift)
            return
        for child in ast.iter_child_nodes(e):
            if isinstance(child, ast.expr):
                self._expr_accesses(child, region, out, env, shift)

    def _elts3(self, name: str, elts):
        """``surf[1, 0]`` / ``prof[1]``: a field with fewer axes takes one offset per axis it has; ``field[K - 1]`` / ``field[I + 1, J]``:
        offsets named by axis, zero on the axes not named."""
        named = {}
        for x in elts:
            ax, rest = None, None
            if isinstance(x, ast.Name) and x.id in ("I", "J", "K"):
                ax, rest = x.id, ast.Constant(value=0)
            elif isinstance(x, ast.BinOp) and isinstance(x.left, ast.Name) and x.left.id in ("I", "J", "K") and isinstance(x.op, (ast.Add, ast.Sub)):
                ax, rest = x.left.id, (x.right if isinstance(x.op, ast.Add) else ast.UnaryOp(op=ast.USub(), operand=x.right))
            if ax is None or ax in named or self._is_scalar_or_field(ax):
                named = None
                break
            named[ax] = rest
        if named:
            zero = ast.Constant(value=0)
            return [named.get(ax, zero) for ax in "IJK"]
        if len(elts) == 3:
            return elts
        axes = self._axes_of(name)
        if axes is not None and len(axes) == len(elts) and len(axes) < 3:
            by_axis = dict(zip(axes, elts))
            zero = ast.Constant(value=0)
            return [by_axis.get(ax, zero) for ax in "IJK"]
        return None

    def _is_scalar_or_field(self, name: str) -> bool:
        return any(n == name for n, _, _ in self.params)

    def _axes_of(self, name: str):
        if hasattr(self, "flds") and name in self.flds:
            return self.flds[name].axes
        for n, axes, _ in self.params:
            if n == name:
                return axes
        typed = getattr(self, "_typed", {}).get(name)
        return typed[1] if typed else None

    def _is_field(self, name: str) -> bool:
        return name in self.flds or name in self._temp_names()

    def _field_data_dims(self, name):
        f = self.flds.get(name)
        return f.data_dims if f is not None else ()

    def _temp_names(self):
        if not hasattr(self, "_temps"):
            names = set()
            scal = {p[0] for p in self.params if p[1] is None}
            api = {p[0] for p in self.params if p[1] is not None}
            for b in self.blocks:
                for n in ast.walk(ast.Module(body=b.stmts, type_ignores=[])):
                    tgt = []
                    if isinstance(n, ast.Assign):
                        tgt = n.targets
                    elif isinstance(n, (ast.AugAssign, ast.AnnAssign)):
                        tgt = [n.target]
                    for t in tgt:
                        for x in (t.elts if isinstance(t, ast.Tuple) else [t]):
                            while isinstance(x, ast.Subscript):
                                x = x.value
                            if isinstance(x, ast.Name) and x.id not in api and x.id not in scal:
                                names.add(x.id)
            self._temps = names
        return self._temps

    def _region(self, node: ast.With):
        if len(node.items) != 1:
            raise Unsupported("several regions")
        call = node.items[0].context_expr
        if not (isinstance(call, ast.Call) and getattr(call.func, "id", None) == "horizontal" and len(call.args) == 1):
            raise Unsupported("with inside an interval")
        sub = call.args[0]
        if not (isinstance(sub, ast.Subscript) and getattr(sub.value, "id", None) == "region"):
            raise Unsupported("horizontal() argument")
        elts = self._subscript_elts(sub)
        if len(elts) != 2:
            raise Unsupported("region rank")
        return tuple(self._region_axis(x, ax) for x, ax in zip(elts, "IJ"))

    def _axis_index(self, node, axis) -> _Bound:
        """``I[0]`` / ``I[-1]`` (+- n): the first / last point of the compute domain."""
        if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub)):
            b = self._axis_index(node.left, axis)
            n = self._const_int(node.right)
            return _Bound(b.end, b.offset + (n if isinstance(node.op, ast.Add) else -n))
        if isinstance(node, ast.Subscript) and getattr(node.value, "id", None) == axis:
            n = self._const_int(node.slice)
            return _Bound(n < 0, n)
        raise Unsupported(f"region bound {ast.unparse(node)}")

    def _region_axis(self, node, axis):
        """(start bound or None, end bound or None), end exclusive."""
        if isinstance(node, ast.Slice):
            lo = self._axis_index(node.lower, axis) if node.lower is not None else None
            hi = self._axis_index(node.upper, axis) if node.upper is not None else None
            return lo, hi
        b = self._axis_index(node, axis)
        return b, _Bound(b.end, b.offset + 1)

    @staticmethod
    def _overlap_along_axis(extent, interval):
        """gtc/passes/horizontal_masks.py:15-47."""
        lo, hi = interval
        if lo is None:
            start_diff = 1000
        elif not lo.end:
            start_diff = extent[0] - lo.offset
        else:
            start_diff = None
        if hi is None:
            end_diff = -1000
        elif hi.end:
            end_diff = extent[1] - hi.offset
        else:
            end_diff = None
        if start_diff is not None and start_diff > 0 and end_diff is None and hi is not None:
            if hi.offset <= extent[0]:
                return None
        elif end_diff is not None and end_diff < 0 and start_diff is None and lo is not None:
            if lo.offset > extent[1]:
                return None
        start_diff = min(start_diff, 0) if start_diff is not None else -10000
        end_diff = max(end_diff, 0) if end_diff is not None else 10000
        return start_diff, end_diff

    def _access_extent(self, block: Extent, off, region) -> Optional[Extent]:
        if region is None:
            return tuple((lo + o, hi + o) for (lo, hi), o in zip(block, off))  # type: ignore[return-value]
        dist = [self._overlap_along_axis(ext, iv) for ext, iv in zip(block, region)]
        if any(d is None for d in dist):
            return None
        ext = tuple((lo - d[0] + o, hi - d[1] + o) for (lo, hi), d, o in zip(block, dist, off))
        return _union(ext, _ZERO)  # type: ignore[arg-type]

    def _extents(self):
        n = len(self._hexecs)
        acc = [self._accesses(s) for _, s in self._hexecs]
        need: Dict[str, Extent] = {}
        self.block_extent: List[Extent] = [_ZERO] * n
        for h in range(n - 1, -1, -1):
            ext = _ZERO
            for name, _, is_write, _ in acc[h]:
                if is_write:
                    ext = _union(ext, need.setdefault(name, _ZERO))
            self.block_extent[h] = ext
            for name, off, is_write, region in acc[h]:
                if is_write:
                    continue
                e = self._access_extent(ext, off, region)
                if e is not None:
                    need[name] = _union(need[name], e) if name in need else e
        self.field_extent = need
        self.written_api = {name for a in acc for name, _, w, _ in a if w and name in self.flds}
        # the reference's order (merged adjacent blocks of one loop order are visited FORWARD, statements backward within each)
        groups: List[List[int]] = []
        for b, block in enumerate(self.blocks):
            prev = self.blocks[b - 1] if b else None
            glued = False
            if prev is not None and prev.order == block.order:
                a_lim, b_lim = (prev.start, block.stop) if block.order == "BACKWARD" else (prev.stop, block.start)
                glued = (a_lim.end, a_lim.offset) == (b_lim.end, b_lim.offset)
            if glued:
                groups[-1].append(b)
            else:
                groups.append([b])
        need2: Dict[str, Extent] = {}
        ext2: List[Extent] = [_ZERO] * n
        for group in reversed(groups):
            for b in group:
                for h in [h for h in range(n - 1, -1, -1) if self._hexecs[h][0] == b]:
                    ext = _ZERO
                    for name, _, is_write, _ in acc[h]:
                        if is_write:
                            ext = _union(ext, need2.setdefault(name, _ZERO))
                    ext2[h] = ext
                    for name, off, is_write, region in acc[h]:
                        if not is_write:
                            e = self._access_extent(ext, off, region)
                            if e is not None:
                                need2[name] = _union(need2[name], e) if name in need2 else e
        self.forward_section_quirk = sorted({name for h in range(n) if ext2[h] != self.block_extent[h]
                                             for name, _, w, _ in acc[h] if w})

    # ---- running ---------------------------------------------------------------------------------------------------------------
    def _k_range(self, block: _Block) -> Tuple[int, int]:
        nk = self.domain[2]
        lo = (nk if block.start.end else 0) + block.start.offset
        hi = (nk if block.stop.end else 0) + block.stop.offset
        if not (0 <= lo <= hi <= nk):
            raise Unsupported(f"interval [{lo}, {hi}) outside 0..{nk}")
        return lo, hi

    def _run_block(self, b_idx: int, block: _Block):
        lo, hi = self._k_range(block)
        hexecs = [(h, s) for h, (b, s) in enumerate(self._hexecs) if b == b_idx]
        if block.order == "PARALLEL":
            for h, s in hexecs:
                self._run_hexec(h, s, lo, hi)
        else:
            levels = range(lo, hi) if block.order == "FORWARD" else range(hi - 1, lo - 1, -1)
            for k in levels:
                for h, s in hexecs:
                    self._run_hexec(h, s, k, k + 1)

    def _run_hexec(self, h, stmt, k0, k1):
        (ilo, ihi), (jlo, jhi) = self.block_extent[h]
        ni, nj, _ = self.domain
        box = (self.H + ilo, self.H + ni + ihi, self.H + jlo, self.H + nj + jhi, self.HK + k0, self.HK + k1)
        self._ext = self.block_extent[h]
        self._exec(stmt, box, ())

    def _mask_now(self, masks, box):
        """The mask of an assignment on ``box``: the AND of the enclosing terms -- ("fixed", array, its box): the mask temporary of
        an ``if``, assigned once; ("lazy", condition): a ``while`` condition in the numpy backend's semantics, evaluated NOW."""
        out = None
        for term in masks:
            if term[0] == "fixed":
                _, arr, abox = term
                if isinstance(arr, np.ndarray) and arr.ndim:
                    full = np.broadcast_to(arr, self._box_shape(abox))
                    arr = full[box[0] - abox[0]: box[1] - abox[0], box[2] - abox[2]: box[3] - abox[2], box[4] - abox[4]: box[5] - abox[4]]
            else:
                arr = self._eval(term[1], box, {}, (0, 0, 0))
            out = arr if out is None else np.logical_and(out, arr)
        return out

    def _exec(self, stmt, box, masks):
        if isinstance(stmt, (ast.Pass, ast.ImportFrom)):  # (`from __externals__ import X` may stand inside a block)
            return
        if isinstance(stmt, ast.With):
            sub = self._region_box(self._region(stmt), box)
            if sub is None:
                return
            for s in stmt.body:
                self._exec(s, sub, masks)
            return
        if isinstance(stmt, ast.If):
            cond = self._eval(stmt.test, box, {}, (0, 0, 0))
            cond = np.array(cond, copy=True) if isinstance(cond, np.ndarray) else cond  # the mask is a temporary: fixed from here on
            for s in stmt.body:
                self._exec(s, box, masks + (("fixed", cond, box),))
            for s in stmt.orelse:
                self._exec(s, box, masks + (("fixed", np.logical_not(cond), box),))
            return
        if isinstance(stmt, ast.While):
            guard = 0
            while True:
                m = self._mask_now(masks + (("lazy", stmt.test),), box)
                if not np.any(m):
                    break
                m = np.array(m, copy=True) if isinstance(m, np.ndarray) else m
                for n_s, s in enumerate(stmt.body):
                    if n_s and not self._while_differs_counted:
                        now = self._mask_now(masks + (("lazy", stmt.test),), box)
                        if not np.array_equal(np.broadcast_to(now, self._box_shape(box)), np.broadcast_to(m, self._box_shape(box))):
                            self.while_reevaluates += 1
                            self._while_differs_counted = True
                    if self.while_semantics == "numpy":
                        self._exec(s, box, masks + (("lazy", stmt.test),))
                    else:
                        self._exec(s, box, (("fixed", m, box),))
                guard += 1
                if guard > 100000:
                    raise Unsupported("while loop does not end")
            return
        mask = self._mask_now(masks, box) if masks else None
        if isinstance(stmt, ast.AugAssign):
            value = ast.BinOp(left=self._as_load(stmt.target), op=stmt.op, right=stmt.value)
            self._assign(stmt.target, value, box, mask, None)
            return
        if isinstance(stmt, ast.AnnAssign):
            self._declare(stmt.target.id, *self._declared(stmt.annotation))
            if stmt.value is not None:
                self._assign(stmt.target, stmt.value, box, mask, None)
            return
        if isinstance(stmt, ast.Assign):
            t = stmt.targets[0]
            if isinstance(t, ast.Tuple):
                fn = self._gt_function(stmt.value) if isinstance(stmt.value, ast.Call) else None
                if fn is not None:
                    rets = self._expand_call(fn, stmt.value, {}, (0, 0))
                    if len(rets) != len(t.elts):
                        raise Unsupported("tuple arity")
                    values = [self._eval(r, box, fenv, (0, 0, 0)) for r, fenv in rets]
                elif isinstance(stmt.value, ast.Tuple) and len(stmt.value.elts) == len(t.elts):
                    values = [self._eval(v, box, {}, (0, 0, 0)) for v in stmt.value.elts]
                else:
                    raise Unsupported("tuple assignment")
                for x, v in zip(t.elts, values):
                    self._assign(x, None, box, mask, v)
                return
            self._assign(t, stmt.value, box, mask, None)
            return
        raise Unsupported(f"statement {type(stmt).__name__}")

    @staticmethod
    def _as_load(t):
        return ast.parse(ast.unparse(t), mode="eval").body

    def _box_shape(self, box):
        return (box[1] - box[0], box[3] - box[2], box[5] - box[4])

    def _region_box(self, region, box):
        """The part of the (extended) box a region covers (oir_to_npir.py:198-207 / horizontal_masks.py:61-112)."""
        out = []
        for (lo, hi), n, b0, b1 in zip(region, self.domain[:2], (box[0], box[2]), (box[1], box[3])):
            a = b0 if lo is None else max(b0, self.H + (n if lo.end else 0) + lo.offset)
            z = b1 if hi is None else min(b1, self.H + (n if hi.end else 0) + hi.offset)
            if a >= z:
                return None
            out += [a, z]
        return (*out, box[4], box[5])

    def _declared(self, ann):
        """``name: Field[IJ, np.float64]`` / ``Field[(np.float32, (2,))]`` -> (dtype, axes, data dimensions)"""
        if not (isinstance(ann, ast.Subscript) and getattr(ann.value, "id", None) == "Field"):
            raise Unsupported("annotated assignment")
        elts = self._subscript_elts(ann)
        axes = ("I", "J", "K")
        if isinstance(elts[0], ast.Name) and elts[0].id in _AXES_NAMES:
            axes = _AXES_NAMES[elts[0].id]
            elts = elts[1:]
        spec, dims = elts[0], ()
        if isinstance(spec, ast.Tuple):
            spec, dims = spec.elts[0], tuple(self._const_int(x) for x in spec.elts[1].elts)
        elif len(elts) == 2 and isinstance(elts[1], ast.Tuple):  # Field[(dtype, (2, 2))] parses like Field[dtype, (2, 2)]
            dims = tuple(self._const_int(x) for x in elts[1].elts)
        txt = ast.unparse(spec)
        if isinstance(spec, ast.Name) and isinstance(self.globals.get(spec.id), type) and issubclass(self.globals[spec.id], np.generic):
            return np.dtype(self.globals[spec.id]), axes, dims  # an alias (F8 = np.float64)
        for key, dt in (("float32", np.float32), ("float64", np.float64), ("int32", np.int32), ("int64", np.int64), ("int8", np.int8),
                        ("int16", np.int16), ("bool", np.bool_)):
            if key in txt:
                return np.dtype(dt), axes, dims
        if txt == "float":
            return np.dtype(np.float64), axes, dims
        if txt == "int":
            return np.dtype(np.int64), axes, dims
        raise Unsupported(f"dtype {txt}")

    def _declare(self, name, dtype, axes, data_dims=()):
        if name not in self.flds:
            f = _Field(name, axes, False, dtype, tuple(data_dims))
            f.frame = self._empty(axes, dtype, data_dims)
            self.flds[name] = f

    def _assign(self, target, value_ast, box, mask, value):
        name, t_off, data_index = self._target(target)
        if value is None:
            value = self._eval(value_ast, box, {}, (0, 0, 0))
        if name not in self.flds:
            if name in self.scalars:
                raise Unsupported("assignment to a scalar parameter")
            if name in self._typed:
                self._declare(name, *self._typed[name])
            else:
                self._declare(name, _dtype_of(value), ("I", "J", "K"))
        f = self.flds[name]
        value = _cast(value, f.dtype)
        if not isinstance(t_off[2], int):  # run-time K offset of the write: scatter along K
            if data_index is not None or f.data_dims or "K" not in f.axes:
                raise Unsupported("run-time K offset on this kind of field")
            idx = self._k_index(self._eval(t_off[2], box, {}, (0, 0, 0)), box, relative=True)
            sub = f.frame[self._slices(f, box, (0, 0, 0))[:2] + (slice(None),)]
            if sub.shape[:2] != idx.shape[:2]:
                raise Unsupported("run-time K offset on a field without I / J")
            old = np.take_along_axis(sub, idx, axis=2)
            new = np.broadcast_to(value, idx.shape)
            np.put_along_axis(sub, idx, new if mask is None else np.where(mask, new, old), axis=2)
            return
        sl = self._slices(f, box, t_off)
        if data_index is not None:
            idx = tuple(int(np.asarray(self._eval(x, box, {}, (0, 0, 0)))) for x in data_index)
            sl = sl + idx
        dest = f.frame[sl]
        value = np.broadcast_to(value, self._box_shape(box)) if np.ndim(value) else value
        if dest.shape != self._box_shape(box):  # a field without some axis: the value must not vary along it
            value = self._collapse(value, dest.shape, mask, box)
            if mask is not None and isinstance(mask, np.ndarray) and mask.ndim:
                mask = self._collapse(np.broadcast_to(mask, self._box_shape(box)), dest.shape, None, box)
        if mask is None:
            f.frame[sl] = value
        else:
            f.frame[sl] = np.where(mask, value, dest)

    @staticmethod
    def _collapse(value, shape, mask, box):
        if not np.ndim(value):
            return value
        idx = tuple(slice(None) if s > 1 else slice(0, 1) for s in shape)
        return value[idx] if value.shape != shape else value

    def _slices(self, f: _Field, box, off):
        di, dj, dk = off
        out = []
        for ax, lo, hi, d in (("I", box[0], box[1], di), ("J", box[2], box[3], dj), ("K", box[4], box[5], dk)):
            if ax in f.axes:
                if lo + d < 0 or hi + d > f.frame.shape["IJK".index(ax)]:
                    raise Unsupported(f"{f.name}: access leaves the frame")
                out.append(slice(lo + d, hi + d))
            else:
                out.append(slice(0, 1))
        return tuple(out)

    # ---- expressions ---------------------------------------------------------------------------------------------------------
    def _gt_function(self, call: ast.Call):
        if isinstance(call.func, ast.Name) and call.func.id not in _NATIVE and call.func.id not in _CASTS:
            obj = self.externals.get(call.func.id, self.globals.get(call.func.id))
            target = getattr(obj, "definition", None) or getattr(obj, "_gtscript_", {}).get("definition") if obj is not None else None
            if target is None and callable(obj) and hasattr(obj, "__code__"):
                target = obj
            if target is not None:
                try:
                    src = textwrap.dedent(inspect.getsource(target))
                except (OSError, TypeError):
                    return None
                fdef = next((n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef)), None)
                return fdef
        return None

    def _expand_call(self, fdef: ast.FunctionDef, call: ast.Call, env, shift):
        """[(returned expression, environment)]: parameters and locals of the function bound to caller expressions."""
        names = [a.arg for a in fdef.args.args]
        fenv: Dict[str, Any] = {}
        for n, a in zip(names, call.args):
            fenv[n] = (a, env, shift)
        for kw in call.keywords:
            fenv[kw.arg] = (kw.value, env, shift)
        defaults = fdef.args.defaults
        for n, d in zip(names[len(names) - len(defaults):], defaults):
            fenv.setdefault(n, (d, {}, (0, 0)))
        rets = None
        for s in fdef.body:
            if isinstance(s, ast.Expr) and isinstance(s.value, ast.Constant):
                continue
            if isinstance(s, ast.ImportFrom):
                continue
            if isinstance(s, ast.Assign) and len(s.targets) == 1 and isinstance(s.targets[0], ast.Name):
                fenv = dict(fenv)
                fenv[s.targets[0].id] = (s.value, dict(fenv), (0, 0))
            elif (isinstance(s, ast.Assign) and len(s.targets) == 1 and isinstance(s.targets[0], ast.Tuple) and isinstance(s.value, ast.Call)
                  and self._gt_function(s.value) is not None):
                inner = self._expand_call(self._gt_function(s.value), s.value, fenv, (0, 0))
                if len(inner) != len(s.targets[0].elts):
                    raise Unsupported("tuple arity")
                fenv = dict(fenv)
                for t, (r_expr, r_env) in zip(s.targets[0].elts, inner):
                    fenv[t.id] = (r_expr, r_env, (0, 0))
            elif isinstance(s, ast.Return):
                v = s.value
                rets = [(x, fenv) for x in v.elts] if isinstance(v, ast.Tuple) else [(v, fenv)]
            else:
                raise Unsupported(f"{type(s).__name__} inside a gtscript function")
        if rets is None:
            raise Unsupported("gtscript function without return")
        return rets

    def _literal(self, v):
        if isinstance(v, (bool, np.bool_)):
            return np.bool_(v)
        if isinstance(v, (int, np.integer)):
            return self.int_t.type(v)
        if isinstance(v, (float, np.floating)):
            return self.float_t.type(v)
        raise Unsupported(f"literal {v!r}")

    def _eval(self, e, box, env, shift):
        """Value of an expression on the box: an array broadcastable to the box's shape or a numpy scalar."""
        if isinstance(e, ast.Constant):
            return self._literal(e.value)
        if isinstance(e, ast.Name):
            if e.id in env:
                sub, sub_env, sub_shift = env[e.id]
                return self._eval(sub, box, sub_env, (shift[0] + sub_shift[0], shift[1] + sub_shift[1], shift[2]))
            if e.id in self.flds:
                return self._read(e.id, box, shift, None)
            if e.id in self.scalars:
                return self.scalars[e.id]
            if e.id in self.externals:
                return self._literal(self.externals[e.id])
            if e.id in ("True", "False"):
                return np.bool_(e.id == "True")
            if e.id == "K":  # the level index (gtscript_frontend.py:1313-1316: an integer of the literal precision)
                return np.arange(box[4] - self.HK, box[5] - self.HK, dtype=self.int_t).reshape(1, 1, -1) + self.int_t.type(shift[2])
            if e.id in self.globals and isinstance(self.globals[e.id], (int, float, bool, np.generic)):
                return self._literal(self.globals[e.id])
            raise Unsupported(f"name {e.id}")
        if isinstance(e, ast.Subscript):
            base = e.value
            if isinstance(base, ast.Subscript):  # field[di, dj, dk][data index]
                idx = tuple(int(np.asarray(self._eval(x, box, env, shift))) for x in self._subscript_elts(e))
                return self._eval_access(base, box, env, shift, idx)
            if isinstance(base, ast.Attribute) and base.attr == "A" and isinstance(base.value, ast.Name) and base.value.id in self.flds:
                # field.A[data index]: the data dimensions at the point itself (gtscript_frontend.py `.A` absolute data indexing)
                idx = tuple(int(np.asarray(self._eval(x, box, env, shift))) for x in self._subscript_elts(e))
                return self._read(base.value.id, box, shift, idx)
            return self._eval_access(e, box, env, shift, None)
        if isinstance(e, ast.UnaryOp):
            v = self._eval(e.operand, box, env, shift)
            if isinstance(e.op, ast.Not):
                return np.logical_not(_cast(v, _ufunc_targets(np.logical_not, [_dtype_of(v)])[0]))
            uf = np.negative if isinstance(e.op, ast.USub) else np.positive
            if isinstance(e.operand, ast.Constant) and isinstance(e.op, ast.USub):
                return self._literal(-e.operand.value)  # "-1.5" is a literal
            (t,) = _ufunc_targets(uf, [_dtype_of(v)])
            return uf(_cast(v, t))
        if isinstance(e, ast.BinOp):
            left, right = self._eval(e.left, box, env, shift), self._eval(e.right, box, env, shift)
            if isinstance(e.op, ast.Pow):
                return np.power(left, right)
            if isinstance(e.op, ast.Mod):
                tl, tr = _ufunc_targets(np.remainder, [_dtype_of(left), _dtype_of(right)])
                return np.remainder(_cast(left, tl), _cast(right, tr))
            uf = _BIN_UFUNC.get(type(e.op))
            if uf is None:
                raise Unsupported(f"operator {type(e.op).__name__}")
            tl, tr = _ufunc_targets(uf, [_dtype_of(left), _dtype_of(right)])
            return uf(_cast(left, tl), _cast(right, tr))
        if isinstance(e, ast.BoolOp):
            uf = np.logical_and if isinstance(e.op, ast.And) else np.logical_or
            acc = self._eval(e.values[0], box, env, shift)
            for v in e.values[1:]:
                r = self._eval(v, box, env, shift)
                tl, tr = _ufunc_targets(uf, [_dtype_of(acc), _dtype_of(r)])
                acc = uf(_cast(acc, tl), _cast(r, tr))
            return acc
        if isinstance(e, ast.Compare):
            if len(e.ops) != 1:
                raise Unsupported("chained comparison")
            left, right = self._eval(e.left, box, env, shift), self._eval(e.comparators[0], box, env, shift)
            uf = _CMP_UFUNC[type(e.ops[0])]
            tl, tr = _ufunc_targets(uf, [_dtype_of(left), _dtype_of(right)])
            return uf(_cast(left, tl), _cast(right, tr))
        if isinstance(e, ast.IfExp):
            cond = self._eval(e.test, box, env, shift)
            a, b = self._eval(e.body, box, env, shift), self._eval(e.orelse, box, env, shift)
            t = max((_dtype_of(a), _dtype_of(b)), key=lambda d: _RANK[d])
            out = np.where(cond, _cast(a, t), _cast(b, t))
            return out if out.ndim else out[()]
        if isinstance(e, ast.Call):
            fn = self._gt_function(e)
            if fn is not None:
                rets = self._expand_call(fn, e, env, shift[:2])
                if len(rets) != 1:
                    raise Unsupported("tuple-valued function in an expression")
                return self._eval(rets[0][0], box, rets[0][1], (0, 0, shift[2]))
            if isinstance(e.func, ast.Attribute) and e.func.attr == "at" and isinstance(e.func.value, ast.Name):
                # field.at(K=expr): the level with that index, counted from the field's origin (gtc/debug/debug_codegen.py:300-312)
                if len(e.keywords) != 1 or e.keywords[0].arg != "K" or e.args:
                    raise Unsupported(f".at form {ast.unparse(e)}")
                return self._gather(e.func.value.id, box, shift, self._eval(e.keywords[0].value, box, env, shift), relative=False, data_index=None)
            if not isinstance(e.func, ast.Name):
                raise Unsupported(f"call {ast.unparse(e.func)}")
            args = [self._eval(a, box, env, shift) for a in e.args]
            if e.func.id in _CASTS:
                return _cast(args[0], np.dtype(_CASTS[e.func.id]))
            if e.func.id == "float":
                return _cast(args[0], self.float_t)
            if e.func.id == "int":
                return _cast(args[0], self.int_t)
            uf = _NATIVE.get(e.func.id)
            if uf is None:
                raise Unsupported(f"function {e.func.id}")
            targets = _ufunc_targets(uf, [_dtype_of(a) for a in args])
            return uf(*[_cast(a, t) for a, t in zip(args, targets)])
        if isinstance(e, ast.Attribute):  # a constant of the enclosing scope (an enum member, a namespace attribute)
            try:
                obj = eval(compile(ast.Expression(body=e), "<const>", "eval"), {"__builtins__": {}}, {**self.globals, **self.externals})
            except Exception as ex:  # noqa: BLE001
                raise Unsupported(f"attribute {ast.unparse(e)}: {ex}") from None
            if isinstance(obj, (bool, int, float, np.generic)):
                return self._literal(int(obj) if isinstance(obj, int) and not isinstance(obj, bool) else obj)
            raise Unsupported(f"attribute {ast.unparse(e)}")
        raise Unsupported(f"expression {type(e).__name__}")

    def _eval_access(self, e: ast.Subscript, box, env, shift, data_index):
        base = e.value
        if not isinstance(base, ast.Name):
            raise Unsupported(f"subscript {ast.unparse(e)}")
        elts = self._subscript_elts(e)
        elts = self._elts3(base.id, elts) or elts
        if base.id in self.flds and len(elts) != 3 and len(elts) == len(self.flds[base.id].data_dims) and data_index is None:
            idx = tuple(int(np.asarray(self._eval(x, box, env, shift))) for x in elts)
            return self._read(base.id, box, shift, idx)
        if len(elts) != 3:
            raise Unsupported(f"offset form {ast.unparse(e)}")
        try:
            dk = self._const_int(elts[2])
        except Unsupported:  # a run-time K offset
            if base.id not in self.flds:
                raise Unsupported(f"run-time K offset on {base.id}")
            return self._gather(base.id, box, (shift[0] + self._const_int(elts[0]), shift[1] + self._const_int(elts[1]), shift[2]),
                                self._eval(elts[2], box, env, shift), relative=True, data_index=data_index)
        off = (self._const_int(elts[0]), self._const_int(elts[1]), dk)
        total = (shift[0] + off[0], shift[1] + off[1], shift[2] + off[2])
        if base.id in env:
            sub, sub_env, sub_shift = env[base.id]
            return self._eval(sub, box, sub_env, (total[0] + sub_shift[0], total[1] + sub_shift[1], total[2]))
        if base.id in self.flds:
            return self._read(base.id, box, total, data_index)
        raise Unsupported(f"offset on {base.id}")

    def _k_index(self, value, box, relative: bool):
        """Frame K indices (an array of the box's shape) for a run-time K offset (relative to each level) or absolute index."""
        shape = self._box_shape(box)
        idx = np.broadcast_to(np.asarray(value), shape).astype(np.int64)
        if relative:
            idx = idx + np.arange(box[4], box[5]).reshape(1, 1, -1)
        else:
            idx = idx + self.HK
        return idx

    def _gather(self, name, box, shift, value, relative, data_index):
        f = self.flds.get(name)
        if f is None or "K" not in f.axes:
            raise Unsupported(f"K index on {name}")
        idx = self._k_index(value, box, relative) + (shift[2] if relative else 0)
        if idx.min() < 0 or idx.max() >= f.frame.shape[2]:
            raise Unsupported(f"{name}: run-time K index leaves the frame")
        sl = self._slices(f, box, (shift[0], shift[1], 0))[:2] + (slice(None),)
        if f.data_dims:
            if data_index is None:
                raise Unsupported(f"{name}: vector field read without a data index")
            sl = sl + tuple(data_index)
        sub = f.frame[sl]
        sub = np.broadcast_to(sub, idx.shape[:2] + sub.shape[2:])
        return np.take_along_axis(sub, idx, axis=2)

    def _read(self, name, box, off, data_index):
        f = self.flds[name]
        if len(off) == 2:
            off = (off[0], off[1], 0)
        sl = self._slices(f, box, off)
        if f.data_dims:
            if data_index is None:
                raise Unsupported(f"{name}: vector field read without a data index")
            sl = sl + tuple(data_index)
        return f.frame[sl]


def run(definition, fields, scalars, domain, **kwargs) -> Interpreter:
    it = Interpreter(definition, **kwargs)
    it(fields, scalars, domain)
    return it
