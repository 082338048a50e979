"""Minimal GTScript recogniser: Python ``ast`` of a stencil definition -> typed ``ir.Stencil``.

The reference's frontend (/root/reference/src/gt4py/cartesian/frontend/gtscript_frontend.py, 4.5 kLoC)
is a general compiler front end and out of scope.  This module parses the
sub-language the hot-path stencils and their known-answer tests are written in
and applies the reference's VALUE-relevant rules exactly:

* expression trees as Python parses them (left-assoc), ``-4.0`` = UnaryOp(neg, 4.0)
  (gtscript_frontend.py:1477-1504);
* literals typed by ``literal_int_precision`` / ``literal_float_precision`` (:1236-1266);
* temporaries typed by their first right-hand side (gtc/passes/gtir_dtype_resolver.py:54-57);
* operand promotion by the smallest matching numpy-ufunc signature, ternary branches to the max
  dtype, assignment casts to the target dtype (gtc/passes/gtir_upcaster.py:43-143);
* interval arithmetic and ordering checks (gtscript_frontend.py:133-160, 1037-1065, 1124-1167).

Anything outside the subset raises ``GTScriptSyntaxError`` -- never a silent approximation.
"""

from __future__ import annotations

import ast
import inspect
import numbers
import textwrap
from dataclasses import replace
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import call_inliner, definitions as gt_definitions, gtscript, ir
from .definitions import GTScriptDefinitionError, GTScriptSymbolError, GTScriptSyntaxError

# DataType order of the reference (gtc/common.py:105-118): bool < int8 < ... < float64.
_RANK = {
    np.dtype("bool"): 10,
    np.dtype("int8"): 11,
    np.dtype("int16"): 12,
    np.dtype("int32"): 14,
    np.dtype("int64"): 18,
    np.dtype("float32"): 104,
    np.dtype("float64"): 108,
}
_CHAR_TO_DTYPE = {np.dtype(d).char: np.dtype(d) for d in _RANK}

_BIN_OPS = {
    ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/", ast.Mod: "%", ast.Pow: "**",
}
_CMP_OPS = {ast.Gt: ">", ast.Lt: "<", ast.GtE: ">=", ast.LtE: "<=", ast.Eq: "==", ast.NotEq: "!="}
_OP_UFUNC = {
    "+": np.add, "-": np.subtract, "*": np.multiply, "/": np.true_divide, "%": np.remainder,
    "**": np.power, ">": np.greater, "<": np.less, ">=": np.greater_equal, "<=": np.less_equal,
    "==": np.equal, "!=": np.not_equal, "and": np.logical_and, "or": np.logical_or,
    "neg": np.negative, "pos": np.positive, "not": np.logical_not,
}
_NATIVE_UFUNC = {
    "abs": np.abs, "min": np.minimum, "max": np.maximum, "mod": np.remainder, "sin": np.sin,
    "cos": np.cos, "tan": np.tan, "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan,
    "sinh": np.sinh, "cosh": np.cosh, "tanh": np.tanh, "asinh": np.arcsinh, "acosh": np.arccosh,
    "atanh": np.arctanh, "sqrt": np.sqrt, "exp": np.exp, "log": np.log, "log10": np.log10,
    "cbrt": np.cbrt, "isfinite": np.isfinite, "isinf": np.isinf, "isnan": np.isnan,
    "floor": np.floor, "ceil": np.ceil, "trunc": np.trunc,
}


class _FloatOnly:
    """Signature of the reference's non-ufunc math functions: float32 -> float32, float64 -> float64
    (gtc/ufuncs.py:16-40)."""

    types = ["f->f", "d->d"]

    def __init__(self, name):
        self.__name__ = name


for _name in ("erf", "erfc", "gamma", "round", "round_away_from_zero"):
    _NATIVE_UFUNC[_name] = _FloatOnly(_name)
_CAST_FUNCS = {"int32": np.dtype("int32"), "int64": np.dtype("int64"), "float32": np.dtype("float32"),
               "float64": np.dtype("float64")}


def ufunc_signature(ufunc: np.ufunc, dtypes: Sequence[np.dtype]) -> Tuple[np.dtype, ...]:
    """Input dtypes of the smallest ufunc loop every operand fits in.

    Restates ``_numpy_ufunc_upcasting_rule`` (gtc/passes/gtir_upcaster.py:43-68): among the loops in
    ``ufunc.types`` whose every input is >= the operand (in the reference's DataType order) pick
    the one with the smallest sum of DataType values.
    """
    best_key, best = None, None
    for sig in ufunc.types:
        ins, _ = sig.split("->")
        if len(ins) != len(dtypes) or any(c not in _CHAR_TO_DTYPE for c in ins):
            continue
        cand = tuple(_CHAR_TO_DTYPE[c] for c in ins)
        if all(_RANK[a] <= _RANK[c] for a, c in zip(dtypes, cand)):
            key = sum(_RANK[c] for c in cand)
            if best_key is None or key < best_key:
                best_key, best = key, cand
    if best is None:
        raise GTScriptSyntaxError(f"No numpy loop of '{ufunc.__name__}' accepts operand dtypes {list(map(str, dtypes))}")
    return best


def _max_dtype(*dtypes: np.dtype) -> np.dtype:
    return max(dtypes, key=lambda d: _RANK[d])


class _Parser(ast.NodeVisitor):
    def __init__(self, definition, annotations, externals, options: gt_definitions.BuildOptions, dtypes=None):
        self.definition = definition
        self.externals = dict(externals)
        self.dtypes = dict(dtypes or {})
        self.context = call_inliner._context_of(definition)
        self.int_dtype = np.dtype(gt_definitions.get_integer_type(options.literal_int_precision))
        self.float_dtype = np.dtype(gt_definitions.get_float_type(options.literal_float_precision))
        # `dtypes={float: np.float32}` also retypes the literals (test_gtscript_frontend.py:1666-1686)
        if float in self.dtypes:
            self.float_dtype = np.dtype(self.dtypes[float])
        if int in self.dtypes:
            self.int_dtype = np.dtype(self.dtypes[int])
        self.fields: Dict[str, ir.FieldDecl] = {}
        self.params: Dict[str, ir.ScalarDecl] = {}
        self.temporaries: Dict[str, ir.FieldDecl] = {}
        self.imported: Dict[str, Any] = {}
        self._order: Optional[ir.LoopOrder] = None
        self._region: Optional[ir.Region] = None  # set while the body of `with horizontal(...)` is parsed
        self._loops: Tuple[Tuple[int, ir.Expr], ...] = ()  # enclosing `while` loops of the statement being parsed
        self._loop_count = 0
        # `while` loops: "statementwise" (default) = the reference's numpy backend, which masks EVERY body statement with the
        # re-evaluated condition (oir_to_npir.py:187-196); "pointwise" = its compiled backends' per-point loop
        # (gtcpp_codegen.py:257, debug_codegen.py:138-144).  They differ when a statement of the body falsifies the condition
        # before the body's last statement (tests/test_oracle_independent.py counts 13 such programs among 150 random ones).
        self._pointwise_while = options.backend_opts.get("while_loops", "statementwise") == "pointwise"
        if options.backend_opts.get("while_loops", "statementwise") not in ("statementwise", "pointwise"):
            raise ValueError(f"Invalid 'while_loops' option ('{options.backend_opts['while_loops']}'): 'statementwise' or 'pointwise'")
        self._groups = 0  # top-level `if` statements seen (one horizontal execution each)
        self._masks = 0
        for pname, ann in annotations.items():
            if isinstance(ann, gtscript._FieldDescriptor):
                axes = tuple(a.name if isinstance(a, gtscript.Axis) else str(a) for a in ann.axes)
                self.fields[pname] = ir.FieldDecl(pname, np.dtype(ann.dtype), axes, tuple(ann.data_dims), True)
            elif isinstance(ann, type) and ann in gtscript.ENUM_REGISTER.values():
                # enums are passed as integers (gtscript_frontend.py:2181-2187)
                self.params[pname] = ir.ScalarDecl(pname, self.int_dtype)
            else:
                try:
                    dt = np.dtype(ann)
                except TypeError as ex:
                    raise GTScriptDefinitionError(
                        f"Invalid annotated dtype value for argument '{pname}': {ann}") from ex
                if dt not in _RANK:
                    raise GTScriptDefinitionError(f"Invalid annotated dtype value for argument '{pname}': {ann}")
                self.params[pname] = ir.ScalarDecl(pname, dt)

    # ---- helpers -------------------------------------------------------------------------
    def _err(self, node, msg):
        line = getattr(node, "lineno", "?")
        return GTScriptSyntaxError(f"{msg} (in '{self.definition.__name__}', line {line})", loc=line)

    def _const(self, node) -> Any:
        """Compile-time value of a small expression (ints, externals, unary minus, None, Ellipsis)."""
        if isinstance(node, ast.Constant):
            return node.value
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = self._const(node.operand)
            return -v if isinstance(node.op, ast.USub) else +v
        if isinstance(node, ast.Name) and node.id in self.imported:
            return self.imported[node.id]
        if isinstance(node, ast.BinOp) and type(node.op) in _BIN_OPS:
            a, b = self._const(node.left), self._const(node.right)
            return eval(f"a {_BIN_OPS[type(node.op)]} b", {}, {"a": a, "b": b})  # noqa: S307 - ints/floats only
        if isinstance(node, ast.Compare) and len(node.ops) == 1 and type(node.ops[0]) in _CMP_OPS:
            a, b = self._const(node.left), self._const(node.comparators[0])
            return eval(f"a {_CMP_OPS[type(node.ops[0])]} b", {}, {"a": a, "b": b})  # noqa: S307
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.Not):
            return not self._const(node.operand)
        raise self._err(node, "Expected a compile-time constant")

    # ---- body structure ------------------------------------------------------------------
    def parse(self) -> ir.Stencil:
        src = textwrap.dedent(inspect.getsource(self.definition))
        tree = ast.parse(src)
        fdef = next(n for n in tree.body if isinstance(n, (ast.FunctionDef,)))
        for imp in call_inliner.inline_calls(fdef, self.definition, self.externals):  # @gtscript.function calls -> statements
            self._visit_import(imp)
        computations: List[ir.Computation] = []
        for stmt in fdef.body:
            if isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant) and isinstance(stmt.value.value, str):
                continue  # docstring
            if isinstance(stmt, ast.ImportFrom):
                self._visit_import(stmt)
                continue
            if isinstance(stmt, ast.Import):
                raise self._err(stmt, f"Invalid 'import' statements ({[a.name for a in stmt.names]})")
            if isinstance(stmt, ast.AnnAssign):
                # typed temporary, optionally initialised: `tmp: Field[np.float32] = 0` at the top of the
                # definition (gtscript_frontend.py:2245-2263); the initial value becomes a PARALLEL
                # full-interval assignment ahead of everything else (:809-850)
                init = self._declare_typed_temporary(stmt)
                if init is not None:
                    computations.append(init)
                continue
            if isinstance(stmt, ast.With):
                computations.extend(self._visit_with(stmt))
                continue
            if isinstance(stmt, ast.Pass):
                continue
            raise self._err(stmt, "Only 'with computation(...)' blocks are allowed at the top level of a stencil")
        if not computations:
            raise self._err(fdef, "Stencil definition contains no computation")
        return ir.Stencil(
            name=self.definition.__name__,
            fields=tuple(self.fields.values()),
            params=tuple(self.params.values()),
            temporaries=tuple(self.temporaries.values()),
            computations=tuple(computations),
        )

    def _declare_annotated(self, node: ast.AnnAssign) -> ir.FieldDecl:
        """Declare the temporary of ``name: <annotation>``.  The annotation is a dtype (``float``, ``np.int32``, a
        key of ``dtypes=``) for a 3-d temporary, ``Field[dtype]`` / ``Field[(dtype, data_dims)]``, or
        ``Field[IJ, dtype]`` for a 2-d temporary (one value per column, kept across K levels and computations;
        the reference supports those on its debug, numpy and dace backends only)."""
        if not isinstance(node.target, ast.Name):
            raise self._err(node, "Only plain names can be annotated as temporaries")
        name = node.target.id
        if name in self.fields or name in self.params or name in self.temporaries:
            raise self._err(node, f"'{name}' is already defined")
        scope = {"Field": gtscript.Field, "np": np, "I": gtscript.I, "J": gtscript.J, "K": gtscript.K,
                 "IJ": gtscript.IJ, "IK": gtscript.IK, "JK": gtscript.JK, "IJK": gtscript.IJK,
                 "int32": np.int32, "int64": np.int64, "float32": np.float32, "float64": np.float64}
        scope.update(getattr(self.definition, "__globals__", {}))
        try:
            closure = inspect.getclosurevars(self.definition)
            scope.update(closure.nonlocals)
        except (TypeError, ValueError):
            pass
        scope.update({k: v for k, v in self.dtypes.items() if isinstance(k, str)})
        source = ast.unparse(node.annotation)
        try:
            descriptor = eval(source, scope)  # noqa: S307 - the user's own annotation
        except Exception as ex:
            raise self._err(node, f"Failed to recognize type {source} for local symbol {name}.") from ex
        descriptor = self.dtypes.get(descriptor, descriptor) if not isinstance(descriptor, gtscript._FieldDescriptor) else descriptor
        if not isinstance(descriptor, gtscript._FieldDescriptor):
            if descriptor is int:
                descriptor = self.int_dtype
            elif descriptor is float:
                descriptor = self.float_dtype
            try:
                descriptor = gtscript._FieldDescriptor(descriptor, gtscript.IJK)
            except (ValueError, TypeError) as ex:
                raise self._err(node, f"Failed to recognize type {source} for local symbol {name}.") from ex
        axes = tuple(a.name if isinstance(a, gtscript.Axis) else str(a) for a in descriptor.axes)
        if axes not in (("I", "J", "K"), ("I", "J")):
            raise self._err(node, f"Typed temporaries must be IJ, temporaries for axes {axes} is not yet available. "
                                  "Contact the team.")
        dtype = descriptor.dtype
        if isinstance(dtype, str):
            if dtype not in self.dtypes:
                raise self._err(node, f"Unknown dtype key '{dtype}' for temporary '{name}' (pass dtypes={{...}})")
            dtype = self.dtypes[dtype]
        dims = tuple(int(n) for n in descriptor.data_dims)
        decl = ir.FieldDecl(name, np.dtype(dtype), axes, dims, False)
        self.temporaries[name] = decl
        return decl

    def _declare_typed_temporary(self, node: ast.AnnAssign) -> Optional[ir.Computation]:
        decl = self._declare_annotated(node)
        name, dims = decl.name, decl.data_dims
        if decl.axes != ("I", "J", "K"):
            del self.temporaries[name]
            raise self._err(node, f"Found {''.join(decl.axes)}, but only IJK is currently supported for temporaries "
                                  "declared at the top of a definition")
        if node.value is None:
            return None
        value = self._const(node.value)
        if not isinstance(value, (bool, numbers.Number)):
            raise self._err(node, "A temporary can only be initialised with a constant")
        import itertools

        group = self._groups
        self._groups += 1
        inits = tuple(ir.Assign(ir.FieldAccess(name, (0, 0, 0), None, None, tuple(index)),
                                self._literal_from_python(value, node), None, group if dims else -1)
                      for index in itertools.product(*(range(n) for n in dims)))
        return ir.Computation(ir.LoopOrder.PARALLEL, (ir.IntervalBlock(ir.Interval.full(), inits),))

    def _visit_import(self, node: ast.ImportFrom) -> None:
        # only `from [gt4py.cartesian.]__externals__ / __gtscript__ import ...` (gtscript_frontend.py:2279-2307)
        roots = ("", "gt4py.cartesian.", "gt4py_amd.cartesian.")
        if node.module in tuple(r + "__gtscript__" for r in roots):
            return
        if node.module not in tuple(r + "__externals__" for r in roots):
            raise self._err(node, f"Invalid 'import' statements (['{node.module}'])")
        for alias in node.names:
            if alias.name not in self.externals:
                raise GTScriptDefinitionError(f"Missing or invalid value for external symbol {alias.name}")
            self.imported[alias.asname or alias.name] = self.externals[alias.name]

    @staticmethod
    def _call_name(node) -> Optional[str]:
        if isinstance(node, ast.Call):
            f = node.func
            if isinstance(f, ast.Name):
                return f.id
            if isinstance(f, ast.Attribute):
                return f.attr
        return None

    def _visit_with(self, node: ast.With) -> List[ir.Computation]:
        names = [self._call_name(item.context_expr) for item in node.items]
        if names and names[0] == "computation":
            order = self._parse_order(node.items[0].context_expr)
            if len(names) == 2 and names[1] == "interval":
                block = self._parse_interval_block(node.items[1].context_expr, node.body, order)
                return [ir.Computation(order, (block,))]
            if len(names) == 3 and names[1] == "interval" and names[2] == "horizontal":
                # `with computation(...), interval(...), horizontal(region[...]):` (gtscript_frontend.py:1195-1199)
                inner = ast.copy_location(ast.With(items=[node.items[2]], body=node.body), node)
                block = self._parse_interval_block(node.items[1].context_expr, [inner], order)
                return [ir.Computation(order, (block,))]
            if len(names) == 1:
                blocks = []
                for inner in node.body:
                    if not (isinstance(inner, ast.With) and len(inner.items) == 1
                            and self._call_name(inner.items[0].context_expr) == "interval"):
                        raise self._err(inner, "Expected 'with interval(...)' inside 'with computation(...)'")
                    blocks.append(self._parse_interval_block(inner.items[0].context_expr, inner.body, order))
                self._check_interval_order(node, order, blocks)
                return [ir.Computation(order, tuple(blocks))]
        raise self._err(node, "Invalid 'with' statement: expected 'with computation(ORDER), interval(...)'")

    def _parse_order(self, call: ast.Call) -> ir.LoopOrder:
        if len(call.args) != 1:
            raise self._err(call, "computation() takes exactly one iteration order")
        arg = call.args[0]
        name = arg.id if isinstance(arg, ast.Name) else (arg.attr if isinstance(arg, ast.Attribute) else None)
        try:
            return {"PARALLEL": ir.LoopOrder.PARALLEL, "FORWARD": ir.LoopOrder.FORWARD,
                    "BACKWARD": ir.LoopOrder.BACKWARD}[name]
        except KeyError:
            raise self._err(call, "Invalid iteration order, expected PARALLEL, FORWARD or BACKWARD") from None

    def _interval_bound(self, node, call):
        """One bound of ``interval(a, b)``: an integer, None, an external integer or ``K[n] + m`` AxisIndex
        (IntervalParser, gtscript_frontend.py:105-224).  Run-time bounds (a scalar parameter, an IJ field) are the
        reference's experimental feature that only its debug backend implements."""
        error = self._err(call, "Invalid interval range specification")
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id in ("I", "J", "K") \
                and node.value.id not in self.fields:
            raise error  # K[2] written in place: two-argument intervals take plain integers
        runtime = [n for n in ast.walk(node) if isinstance(n, ast.Name) and (n.id in self.fields or n.id in self.params
                                                                            or n.id in self.temporaries)]
        if runtime:
            if isinstance(node, ast.Subscript):
                elts = node.slice.elts if isinstance(node.slice, ast.Tuple) else [node.slice]
                inner = node.value if isinstance(node.value, ast.Subscript) else node
                elts = inner.slice.elts if isinstance(inner.slice, ast.Tuple) else [inner.slice]
                if any(not (isinstance(e, ast.Constant) and e.value == 0) for e in elts):
                    raise error  # an index field read at an offset
            raise NotImplementedError("Runtime interval bounds not implemented yet.")
        try:
            value = self._const(node)
        except (GTScriptSyntaxError, GTScriptSymbolError):
            raise error from None
        if isinstance(value, gtscript.AxisIndex):
            if value.axis != "K":
                raise error
            return ir.AxisBound(ir.Level.START if value.index >= 0 else ir.Level.END, value.index + value.offset)
        if value is None or (isinstance(value, numbers.Integral) and not isinstance(value, (bool, np.bool_))):
            return None if value is None else int(value)
        raise error

    def _parse_interval(self, call: ast.Call) -> ir.Interval:
        args = list(call.args) or [kw.value for kw in call.keywords]
        if len(args) == 1 and isinstance(args[0], ast.Constant) and args[0].value is Ellipsis:
            return ir.Interval.full()
        if len(args) != 2:
            raise self._err(call, "Invalid interval range specification: expected interval(...) or interval(start, end)")
        lo, hi = self._interval_bound(args[0], call), self._interval_bound(args[1], call)
        if lo is None:
            raise self._err(call, "Invalid interval range specification")
        if isinstance(lo, ir.AxisBound):
            start = lo
        else:
            start = ir.AxisBound(ir.Level.START, lo) if lo >= 0 else ir.AxisBound(ir.Level.END, lo)
        if isinstance(hi, ir.AxisBound):
            end = hi
        elif hi is None:
            end = ir.AxisBound(ir.Level.END, 0)
        elif hi < 0:
            end = ir.AxisBound(ir.Level.END, hi)
        else:
            end = ir.AxisBound(ir.Level.START, hi)
        # reject reversed intervals and those empty for every domain size (gtscript_frontend.py:1152-1165)
        if start.level is ir.Level.END and end.level is ir.Level.START:
            raise self._err(call, "Invalid interval range specification")
        if start.level == end.level and start.offset >= end.offset:
            raise self._err(call, "Invalid interval range specification")
        return ir.Interval(start, end)

    def _check_interval_order(self, node, order, blocks: List[ir.IntervalBlock]) -> None:
        """Blocks must be disjoint and listed in execution order (gtscript_frontend.py:1037-1065)."""
        big = 10_000

        def key(b: ir.AxisBound) -> int:
            return b.offset if b.level is ir.Level.START else big + b.offset

        ranges = [(key(b.interval.start), key(b.interval.end)) for b in blocks]
        for n, (a0, a1) in enumerate(ranges):
            for b0, b1 in ranges[n + 1:]:
                if a0 < b1 and b0 < a1:
                    raise self._err(node, "Overlapping intervals detected in computation")
        seq = ranges if order is not ir.LoopOrder.BACKWARD else list(reversed(ranges))
        for (a0, a1), (b0, b1) in zip(seq, seq[1:]):
            if a1 > b0:
                raise self._err(node, "Intervals must be specified in order of execution")

    def _parse_interval_block(self, call, body, order) -> ir.IntervalBlock:
        interval = self._parse_interval(call)
        self._order = order
        stmts: List[ir.Assign] = []
        for stmt in body:
            stmts.extend(self._visit_stmt(stmt))
        return ir.IntervalBlock(interval, tuple(stmts))

    # ---- statements ----------------------------------------------------------------------
    def _visit_stmt(self, node, mask: Optional[ir.Expr] = None, group: int = -1) -> List[ir.Assign]:
        if isinstance(node, ast.Assign):
            if len(node.targets) != 1:
                raise self._err(node, "Chained assignment is not supported")
            return self._make_assign(node.targets[0], self.visit(node.value), node, mask, group)
        if isinstance(node, ast.ImportFrom):  # `from __externals__ import X` inside a computation
            self._visit_import(node)
            return []
        if isinstance(node, ast.AnnAssign):
            # `tmp: float = 0` / `tmp: Field[IJ, float] = 0` inside a computation: declares the temporary on its
            # first assignment (gtscript_frontend.py:1840-1905)
            if not isinstance(node.target, ast.Name) or node.value is None:
                raise self._err(node, "Only initialised plain names can be annotated inside a computation")
            if node.target.id not in self.fields and node.target.id not in self.temporaries:
                self._declare_annotated(node)
            return self._make_assign(node.target, self.visit(node.value), node, mask, group)
        if isinstance(node, ast.AugAssign):
            if type(node.op) not in _BIN_OPS:
                raise self._err(node, "Unsupported augmented assignment")
            target_read = self._target_access(node.target, node, reading=True)
            value = ir.BinaryOp(_BIN_OPS[type(node.op)], target_read, self.visit(node.value))
            return self._make_assign(node.target, value, node, mask, group)
        if isinstance(node, ast.If):
            test = node.test
            out: List[ir.Assign] = []
            if self._call_name(test) == "__INLINED":
                branch = node.body if self._const(test.args[0]) else node.orelse
                for s in branch:
                    out.extend(self._visit_stmt(s, mask, group))
                return out
            # Run-time if: flattened into masked assignments (gtir_to_oir.py:146-218).  A condition that
            # reads fields is evaluated ONCE into a boolean temporary before either branch runs; a
            # scalar condition is used as it is.  Nested masks are AND-ed (oir_to_npir.py visit_MaskStmt).
            if group < 0:
                group = self._groups
                self._groups += 1
            cond = self.visit(test)
            if any(isinstance(e, ir.FieldAccess) and e.data_index is None for e in ir.walk(cond)):
                raise self._err(node, "Conditions must index the data dimensions of the fields they read")
            if any(isinstance(e, ir.FieldAccess) for e in ir.walk(cond)):
                name = f"mask_{self._masks}"
                while name in self.fields or name in self.params or name in self.temporaries:
                    self._masks += 1
                    name = f"mask_{self._masks}"
                self._masks += 1
                self.temporaries[name] = ir.FieldDecl(name, np.dtype("bool"), ("I", "J", "K"), (), False)
                out.append(ir.Assign(ir.FieldAccess(name, (0, 0, 0)), cond, mask, group, self._region, self._loops))
                cond = ir.FieldAccess(name, (0, 0, 0))
            for branch, this in ((node.body, cond), (node.orelse, ir.UnaryOp("not", cond))):
                if not branch:
                    continue
                combined = this if mask is None else ir.BinaryOp("and", mask, this)
                for s in branch:
                    out.extend(self._visit_stmt(s, combined, group))
            return out
        if isinstance(node, ast.While):
            # `while cond: body` -- per point: repeat the body while the condition holds.  The condition is an
            # expression re-evaluated every iteration (no temporary), AND-ed with the enclosing masks; the
            # body statements are masked by it (oir_to_npir.py:176-185, npir_codegen.py:252-267).
            if node.orelse:
                raise self._err(node, "'while ... else' is not supported")
            if group < 0:
                group = self._groups
                self._groups += 1
            cond = self.visit(node.test)
            if any(isinstance(e, ir.FieldAccess) and e.data_index is None for e in ir.walk(cond)):
                raise self._err(node, "Conditions must index the data dimensions of the fields they read")
            full = cond if mask is None else ir.BinaryOp("and", mask, cond)
            saved = self._loops
            self._loops = saved + ((self._loop_count + (ir.POINTWISE_LOOP if self._pointwise_while else 0), full),)
            self._loop_count += 1
            out = []
            try:
                for s in node.body:
                    out.extend(self._visit_stmt(s, mask if self._pointwise_while else full, group))
            finally:
                self._loops = saved
            if not out:
                raise self._err(node, "Empty 'while' body")
            return out
        if isinstance(node, ast.With):
            # `with horizontal(region[...], region[...]):` -- the body is repeated once per region, each
            # copy its own horizontal execution restricted to that region
            # (gtscript_frontend.py:1950-1980; one HorizontalIf per region)
            if len(node.items) != 1 or self._call_name(node.items[0].context_expr) != "horizontal":
                raise self._err(node, "Only 'with horizontal(region[...])' may appear inside an interval block")
            if self._region is not None or any(isinstance(c, ast.With) for c in node.body):
                raise self._err(node, "Cannot nest `with` node inside a horizontal region.")
            call = node.items[0].context_expr
            if not call.args or call.keywords:
                raise self._err(node, "horizontal() takes one or more region[...] arguments")
            out = []
            for arg in call.args:
                self._region = self._parse_region(arg)
                this_group = group
                if this_group < 0:
                    this_group = self._groups
                    self._groups += 1
                try:
                    for s in node.body:
                        out.extend(self._visit_stmt(s, mask, this_group))
                finally:
                    self._region = None
            return out
        if isinstance(node, ast.Pass):
            return []
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant):
            return []
        if isinstance(node, ast.Expr) and self._call_name(node.value) == "compile_assert":
            # evaluated now, from constants and externals only (gtscript_frontend.py:774-806)
            if len(node.value.args) != 1:
                raise self._err(node, "Invalid assertion. Correct syntax: compile_assert(condition)")
            try:
                ok = self._const(node.value.args[0])
            except Exception as ex:
                raise self._err(node, "Evaluation of compile_assert condition failed at the preprocessing step") from ex
            if not isinstance(ok, (bool, np.bool_)):
                raise self._err(node, "Evaluation of compile_assert condition failed at the preprocessing step")
            if not ok:
                raise gt_definitions.GTScriptAssertionError(ast.unparse(node), loc=getattr(node, "lineno", None))
            return []
        raise self._err(node, f"Unsupported statement '{type(node).__name__}' in stencil body")

    # ---- horizontal regions --------------------------------------------------------------
    def _parse_region(self, node) -> ir.Region:
        """``region[<I spec>, <J spec>]`` with specs ``I[0]``, ``I[-1]``, ``I[0] + n``, ``a:b`` or ``:``
        (HorizontalIntervalParser, gtscript_frontend.py:226-300; bounds :133-160)."""
        if not (isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id == "region"):
            raise self._err(node, "Invalid horizontal range specification: expected region[...]")
        spec = node.slice
        if not isinstance(spec, ast.Tuple) or len(spec.elts) != 2:
            raise self._err(node, "Invalid horizontal range specification: region takes an I and a J range")
        return ir.Region(self._parse_axis_interval(spec.elts[0], "I"), self._parse_axis_interval(spec.elts[1], "J"))

    def _parse_axis_interval(self, node, axis: str) -> ir.HorizontalInterval:
        def bound(e) -> Optional[ir.AxisBound]:
            """AxisIndex arithmetic: I[0] -> START+0, I[-1] -> END-1, +- integer constants."""
            if e is None:
                return None
            if isinstance(e, ast.Name):  # an AxisIndex handed in as an external: `i1 = I[0] + 1`
                if e.id not in self.imported:
                    raise GTScriptSymbolError(f"Unknown symbol '{e.id}' in a horizontal range specification")
                value = self.imported[e.id]
                if not isinstance(value, gtscript.AxisIndex) or value.axis != axis:
                    raise self._err(e, f"Invalid horizontal range specification: '{e.id}' is not an index on axis {axis}")
                return ir.AxisBound(ir.Level.START if value.index >= 0 else ir.Level.END, value.index + value.offset)
            if isinstance(e, ast.Subscript) and isinstance(e.slice, ast.Slice):
                raise self._err(e, "Invalid interval range specification")  # the retired I[0:2] spelling
            if isinstance(e, ast.Subscript) and isinstance(e.value, ast.Name):
                if e.value.id != axis:
                    raise self._err(e, f"Invalid horizontal range specification: Expected axis {axis}, got {e.value.id}")
                index = self._const(e.slice)
                if index not in (0, -1):
                    raise self._err(e, f"Invalid horizontal range specification: Expected specification {axis}[0] or {axis}[-1]")
                return ir.AxisBound(ir.Level.START, 0) if index == 0 else ir.AxisBound(ir.Level.END, -1)
            if isinstance(e, ast.BinOp) and isinstance(e.op, (ast.Add, ast.Sub)):
                def number(x):
                    try:
                        v = self._const(x)
                    except (GTScriptSyntaxError, GTScriptSymbolError):
                        return None
                    return int(v) if isinstance(v, numbers.Integral) and not isinstance(v, (bool, np.bool_)) else None

                shift = number(e.right)
                if shift is not None:
                    left = bound(e.left)
                    return ir.AxisBound(left.level, left.offset + (shift if isinstance(e.op, ast.Add) else -shift))
                shift = number(e.left)
                if shift is not None and isinstance(e.op, ast.Add):  # n + <index>
                    right = bound(e.right)
                    return ir.AxisBound(right.level, right.offset + shift)
            raise self._err(e, "Invalid horizontal range specification")

        if isinstance(node, ast.Slice):
            if node.step is not None:
                raise self._err(node, "Invalid horizontal range specification: no step allowed")
            return ir.HorizontalInterval(bound(node.lower), bound(node.upper))
        single = bound(node)
        return ir.HorizontalInterval(single, ir.AxisBound(single.level, single.offset + 1))

    def _data_dims(self, name: str) -> Tuple[int, ...]:
        decl = self.fields.get(name) or self.temporaries.get(name)
        return tuple(decl.data_dims) if decl is not None else ()

    def _split_data_index(self, node: ast.Subscript):
        """``f[i, j, k][d0, d1]`` -> (the ``f[i, j, k]`` node, (d0, d1)); ``f.A[d0, d1]`` -> (the ``f`` name node,
        (d0, d1)); plain subscripts -> (node, None).  Elements that are not compile-time constants are integer
        expressions evaluated at run time (gtscript_frontend.py:1429-1455)."""
        inner = node.value
        if isinstance(inner, ast.Subscript) and isinstance(inner.value, ast.Name) and (
                self._data_dims(inner.value.id) or inner.value.id in self.fields or inner.value.id in self.temporaries):
            name = inner.value.id
        elif isinstance(inner, ast.Attribute) and inner.attr == "A" and isinstance(inner.value, ast.Name):
            name, inner = inner.value.id, inner.value
            if name not in self.fields and name not in self.temporaries:
                raise GTScriptSymbolError(f"Unknown field '{name}' in stencil '{self.definition.__name__}'")
        else:
            return node, None
        dims = self._data_dims(name)
        elts = list(node.slice.elts) if isinstance(node.slice, ast.Tuple) else [node.slice]
        index = []
        for e in elts:
            try:
                v = self._const(e)
            except (GTScriptSyntaxError, GTScriptSymbolError):
                index.append(self.visit(e))
                continue
            if isinstance(v, (bool, np.bool_)) or not isinstance(v, numbers.Integral):
                raise self._err(node, f"Invalid data index for field '{name}': {v!r}")
            index.append(int(v))
        if len(index) != len(dims):
            raise self._err(node, f"Incorrect data index length {len(index)}. Invalid data dimension index. "
                                  f"Field {name} has {len(dims)} data dimensions.")
        if any(isinstance(v, int) and not 0 <= v < n for v, n in zip(index, dims)):
            raise self._err(node, f"Data index out of bounds. Found index {index}, but field {name} has {dims} "
                                  "data-dimensions")
        return inner, tuple(index)

    def _target_access(self, target, node, reading=False) -> ir.FieldAccess:
        data_index: Optional[Tuple[int, ...]] = None
        variable: list = []
        if isinstance(target, ast.Subscript):
            if isinstance(target.value, ast.Attribute) and target.value.attr == "A":
                raise self._err(node, "writing to an GlobalTable ('A' global indexation) is forbidden")
            target, data_index = self._split_data_index(target)
        if isinstance(target, ast.Name):
            name, offset = target.id, (0, 0, 0)
        elif isinstance(target, ast.Subscript) and isinstance(target.value, ast.Name):
            name = target.value.id
            offset = self._parse_offset(target, name, variable)
        else:
            raise self._err(node, "Invalid assignment target")
        if data_index is None and not self._data_dims(name):
            data_index = ()
        if offset[0] != 0 or offset[1] != 0:
            raise self._err(node, "Assignment to non-zero offsets is not supported in IJ")
        if (offset[2] != 0 or variable) and self._order is ir.LoopOrder.PARALLEL:
            raise self._err(node, "Assignment to non-zero offsets in K is not available in PARALLEL. "
                                  "Choose FORWARD or BACKWARD.")
        if name in self.params or name in self.imported:
            raise self._err(node, f"Cannot assign to scalar parameter or external '{name}'")
        decl = self.fields.get(name) or self.temporaries.get(name)
        if decl is not None and not reading and not {"I", "J"} <= set(decl.axes):
            # gtscript_frontend.py:1811-1822
            raise self._err(node, f"Cannot assign to field '{name}' as all parallel axes 'I, J' are not present")
        if reading and name not in self.fields and name not in self.temporaries:
            raise GTScriptSymbolError(f"Unknown symbol '{name}'")
        return ir.FieldAccess(name, offset, None, variable[0] if variable else None, data_index)

    def _make_assign(self, target, value: ir.Expr, node, mask: Optional[ir.Expr] = None, group: int = -1) -> List[ir.Assign]:
        access = self._target_access(target, node)
        if access.name not in self.fields and access.name not in self.temporaries:
            self.temporaries[access.name] = ir.FieldDecl(access.name, None, ("I", "J", "K"), (), False)
            access = ir.FieldAccess(access.name, access.offset, None, access.koffset, ())
        # vector-valued statement: accesses to fields with data dimensions that carry no data index stand for
        # the whole vector / matrix; the statement is unrolled into one assignment per element of the target
        # (defir_to_gtir.py:160-192)
        is_open = lambda e: isinstance(e, ir.FieldAccess) and e.data_index is None  # noqa: E731
        if not is_open(access) and not any(is_open(e) for e in ir.walk(value)):
            if any(getattr(e, "op", None) in ("@", "T") for e in ir.walk(value)):
                raise self._err(node, "'@' and '.T' apply to whole fields with data dimensions")
            return [ir.Assign(access, value, mask, group, self._region, self._loops)]
        def not_indexed(e: ir.FieldAccess):
            # DataDimensionsChecker, defir_to_gtir.py:102-121: only statements whose TARGET is a whole vector are unrolled
            decl = self.fields.get(e.name) or self.temporaries.get(e.name)
            cdims, ddims = [0] * len(decl.axes), ["x"] * len(decl.data_dims)
            return self._err(node, f"Field {e.name} has data dimensions but no data dimensions index is specified. "
                                   f"Use `{e.name}.A{ddims}` or `{e.name}{cdims}{ddims}`.")

        if access.data_index is not None:
            raise not_indexed(next(e for e in ir.walk(value) if is_open(e)))
        for extra in ([mask] if mask is not None else []) + [c for _, c in self._loops]:
            for e in ir.walk(extra):
                if is_open(e):
                    raise not_indexed(e)
        dims = self._data_dims(access.name)
        shape = self._vector_shape(value, node)
        if shape and shape != dims:
            raise self._err(node, f"Assignment dimension mismatch: '{access.name}' has dim = {dims}; rhs has dim {shape}")
        if group < 0:
            group = self._groups
            self._groups += 1
        import itertools

        return [ir.Assign(ir.FieldAccess(access.name, access.offset, None, access.koffset, tuple(index)),
                          self._vector_element(value, tuple(index) if shape else ()), mask, group, self._region,
                          self._loops)
                for index in itertools.product(*(range(n) for n in dims))]

    # ---- vector / matrix expressions (fields with data dimensions used without a data index) ------
    # The reference unrolls them into nested lists of scalar expressions (UnrollVectorExpressions,
    # defir_to_gtir.py:196-299).  Here the same unrolling is expressed as (shape, element(index)).
    def _vector_shape(self, e: ir.Expr, node) -> Tuple[int, ...]:
        if isinstance(e, ir.FieldAccess):
            return self._data_dims(e.name) if e.data_index is None else ()
        if isinstance(e, ir.UnaryOp) and e.op == "T":
            shape = self._vector_shape(e.expr, node)
            if len(shape) != 2:
                raise self._err(node, "'.T' applies to fields with two data dimensions")
            return shape[::-1]
        if isinstance(e, ir.BinaryOp) and e.op == "@":
            left, right = self._vector_shape(e.left, node), self._vector_shape(e.right, node)
            if len(left) != 2 or len(right) != 1 or left[1] != right[0]:
                raise self._err(node, f"'@' multiplies a matrix with a vector; got data dimensions {left} @ {right}")
            return left[:1]
        children = {
            ir.UnaryOp: lambda: (e.expr,), ir.Cast: lambda: (e.expr,), ir.BinaryOp: lambda: (e.left, e.right),
            ir.TernaryOp: lambda: (e.cond, e.true_expr, e.false_expr), ir.NativeCall: lambda: e.args,
        }.get(type(e), lambda: ())()
        shapes = {s for s in (self._vector_shape(c, node) for c in children) if s}
        if len(shapes) > 1:
            raise self._err(node, f"Fields with different data dimensions {sorted(shapes)} in one vector expression")
        return shapes.pop() if shapes else ()

    def _vector_element(self, e: ir.Expr, index: Tuple[int, ...]) -> ir.Expr:
        """The scalar expression for element ``index`` of the vector-valued ``e`` (scalars broadcast)."""
        el = self._vector_element
        if isinstance(e, ir.FieldAccess):
            if e.data_index is None:
                return ir.FieldAccess(e.name, e.offset, e.dtype, e.koffset, index)
            return e
        if isinstance(e, ir.UnaryOp) and e.op == "T":
            return el(e.expr, index[::-1])
        if isinstance(e, ir.BinaryOp) and e.op == "@":
            # row . vector accumulated left to right from the first product (defir_to_gtir.py:265-273)
            (row,) = index
            n = self._vector_shape(e.right, None)[0]
            acc = ir.BinaryOp("*", el(e.left, (row, 0)), el(e.right, (0,)))
            for i in range(1, n):
                acc = ir.BinaryOp("+", acc, ir.BinaryOp("*", el(e.left, (row, i)), el(e.right, (i,))))
            return acc
        pick = lambda c: el(c, index if self._vector_shape(c, None) else ())  # noqa: E731
        if isinstance(e, (ir.UnaryOp, ir.Cast)):
            return replace(e, expr=pick(e.expr))
        if isinstance(e, ir.BinaryOp):
            return replace(e, left=pick(e.left), right=pick(e.right))
        if isinstance(e, ir.TernaryOp):
            return replace(e, cond=pick(e.cond), true_expr=pick(e.true_expr), false_expr=pick(e.false_expr))
        if isinstance(e, ir.NativeCall):
            return replace(e, args=tuple(pick(a) for a in e.args))
        return e

    # ---- expressions ---------------------------------------------------------------------
    def generic_visit(self, node):
        raise self._err(node, f"Unsupported expression '{type(node).__name__}'")

    def visit_Constant(self, node: ast.Constant) -> ir.Expr:
        v = node.value
        if isinstance(v, np.generic):  # a typed constant of the enclosing scope, bound by call_inliner.bind_nonlocals
            return self._literal_from_python(v, node)
        if isinstance(v, bool):
            return ir.Literal(v, np.dtype("bool"))
        if isinstance(v, int):
            return ir.Literal(v, self.int_dtype)
        if isinstance(v, float):
            return ir.Literal(v, self.float_dtype)
        raise self._err(node, f"Unsupported literal {v!r}")

    def _literal_from_python(self, value, node, name: Optional[str] = None) -> ir.Expr:
        if isinstance(value, (bool, np.bool_)):
            return ir.Literal(bool(value), np.dtype("bool"))
        if isinstance(value, np.generic) and np.dtype(type(value)) in _RANK:
            return ir.Literal(value.item(), np.dtype(type(value)))
        if isinstance(value, numbers.Integral):
            return ir.Literal(int(value), self.int_dtype)
        if isinstance(value, numbers.Real):
            return ir.Literal(float(value), self.float_dtype)
        raise GTScriptDefinitionError(f"Missing or invalid value for external symbol {name or ''}: {value!r} is not a "
                                      f"supported constant (in '{self.definition.__name__}')")

    def visit_Name(self, node: ast.Name) -> ir.Expr:
        name = node.id
        if name in self.fields or name in self.temporaries:
            return ir.FieldAccess(name, (0, 0, 0), None, None, None if self._data_dims(name) else ())
        if name in self.params:
            return ir.ScalarAccess(name, self.params[name].dtype)
        if name in self.imported:
            return self._literal_from_python(self.imported[name], node, name)
        if name in ("True", "False"):
            return ir.Literal(name == "True", np.dtype("bool"))
        if name == "K":  # the iteration index as a value (gtscript_frontend.py:874-886, 1312-1316)
            return ir.AxisIndex("K", self.int_dtype)
        if name in ("I", "J"):
            raise self._err(node, f"Parallel axis {name} can't be queried - only K")
        raise GTScriptSymbolError(f"Unknown symbol '{name}' in stencil '{self.definition.__name__}'")

    def _parse_offset(self, node: ast.Subscript, name: str, variable: Optional[list] = None) -> Tuple[int, int, int]:
        decl = self.fields.get(name) or self.temporaries.get(name)
        axes = decl.axes if decl is not None else ("I", "J", "K")
        index = node.slice
        elts = list(index.elts) if isinstance(index, ast.Tuple) else [index]
        offset = {"I": 0, "J": 0, "K": 0}
        if any(isinstance(e, ast.Constant) and e.value is Ellipsis for e in elts):
            return 0, 0, 0  # field[...]: the point itself (gtscript_frontend.py:1381-1382)
        if any(isinstance(e, ast.Slice) for e in elts):
            raise self._err(node, "Invalid target in assignment.")

        def axis_of(e) -> Optional[str]:
            if isinstance(e, ast.Name):
                if e.id in ("I", "J", "K") and e.id not in self.imported:
                    return e.id
                value = self.imported.get(e.id, self.context.get(e.id) if e.id not in self.fields else None)
                if isinstance(value, gtscript.Axis):  # an axis handed in as an external
                    return value.name
            return None

        def axis_shift(e) -> Optional[Tuple[str, int]]:
            if axis_of(e) is not None:
                return axis_of(e), 0
            if isinstance(e, ast.BinOp) and any(axis_of(n) for n in ast.walk(e) if isinstance(n, ast.Name)):
                # <axis> +- <integer constant>, nothing else (gtscript_frontend.py:303-373)
                if not isinstance(e.op, (ast.Add, ast.Sub)) or axis_of(e.left) is None:
                    raise self._err(node, "Invalid axis offset: expected <axis> + <integer> or <axis> - <integer>")
                if any(axis_of(n) for n in ast.walk(e.right) if isinstance(n, ast.Name)):
                    raise self._err(node, "Invalid axis offset: an axis may appear only once")
                shift = self._const(e.right)
                if not isinstance(shift, numbers.Integral):
                    raise self._err(node, "Axis offsets must be integer constants")
                return axis_of(e.left), int(shift) if isinstance(e.op, ast.Add) else -int(shift)
            return None

        shifted = [axis_shift(e) for e in elts]
        if all(s is not None for s in shifted):  # new style: f[I + 1], f[J - 1, K + 1]
            seen = []
            for ax, sh in shifted:
                if ax in seen or (seen and "IJK".index(ax) < "IJK".index(seen[-1])):
                    raise self._err(node, "Axis offsets must be given once, in I, J, K order")
                if ax not in axes:
                    raise self._err(node, f"Field '{name}' has no axis {ax}")
                seen.append(ax)
                offset[ax] = sh
        elif any(s is not None for s in shifted):
            raise self._err(node, "Cannot mix axis offsets and integer offsets")
        else:
            values = []
            for pos, e in enumerate(elts):
                try:
                    values.append(self._const(e))
                except (GTScriptSyntaxError, GTScriptSymbolError):
                    # a run-time K index: field[0, 0, <int expression>] (VariableKOffset)
                    if variable is None or len(elts) != len(axes) or axes[pos] != "K":
                        raise
                    variable.append(self.visit(e))
                    values.append(0)
            if len(values) != len(axes):
                hint = f" Did you mean absolute indexing via .A{values}?" if not axes else ""
                raise self._err(node, f"Incorrect offset specification detected for field '{name}'. "
                                      f"Found {values} but the field has dimensions ({', '.join(axes)}).{hint}")
            for ax, v in zip(axes, values):
                if not isinstance(v, numbers.Integral):
                    raise self._err(node, "Field offsets must be integer constants")
                offset[ax] = int(v)
        return offset["I"], offset["J"], offset["K"]

    def visit_Subscript(self, node: ast.Subscript) -> ir.Expr:
        node, data_index = self._split_data_index(node)
        if isinstance(node, ast.Name):  # name.A[...]: centred access with a data index
            return ir.FieldAccess(node.id, (0, 0, 0), None, None, data_index)
        if not isinstance(node.value, ast.Name):
            raise self._err(node, "Only fields can be subscripted")
        name = node.value.id
        if name not in self.fields and name not in self.temporaries:
            raise GTScriptSymbolError(f"Unknown field '{name}' in stencil '{self.definition.__name__}'")
        variable: list = []
        offset = self._parse_offset(node, name, variable)
        if data_index is None and not self._data_dims(name):
            data_index = ()
        return ir.FieldAccess(name, offset, None, variable[0] if variable else None, data_index)

    def visit_UnaryOp(self, node: ast.UnaryOp) -> ir.Expr:
        op = {ast.USub: "-", ast.UAdd: "+", ast.Not: "not"}.get(type(node.op))
        if op is None:
            raise self._err(node, "Unsupported unary operator")
        return ir.UnaryOp(op, self.visit(node.operand))

    def visit_Attribute(self, node: ast.Attribute) -> ir.Expr:
        if node.attr == "T":  # matrix transpose; removed again when the vector statement is unrolled
            return ir.UnaryOp("T", self.visit(node.value))
        if isinstance(node.value, ast.Name) and node.value.id in gtscript.ENUM_REGISTER:
            # MyEnum.A is its integer value (gtscript_frontend.py:449-459)
            return ir.Literal(int(getattr(gtscript.ENUM_REGISTER[node.value.id], node.attr)), self.int_dtype)
        dotted = call_inliner._dotted(node)
        if dotted is not None and dotted.split(".")[0] in self.imported:  # NAMESPACE.A with an imported external
            value = self.imported[dotted.split(".")[0]]
            try:
                for attr in dotted.split(".")[1:]:
                    value = getattr(value, attr)
            except AttributeError as ex:
                raise GTScriptDefinitionError(f"Missing or invalid value for external symbol {dotted}") from ex
            return self._literal_from_python(value, node, dotted)
        raise self._err(node, f"Unsupported attribute access '.{node.attr}'")

    def visit_BinOp(self, node: ast.BinOp) -> ir.Expr:
        if isinstance(node.op, ast.MatMult):  # only inside vector statements, see _vector_element
            return ir.BinaryOp("@", self.visit(node.left), self.visit(node.right))
        if type(node.op) not in _BIN_OPS:
            raise self._err(node, f"Unsupported binary operator '{type(node.op).__name__}'")
        return ir.BinaryOp(_BIN_OPS[type(node.op)], self.visit(node.left), self.visit(node.right))

    def visit_Compare(self, node: ast.Compare) -> ir.Expr:
        if len(node.ops) != 1 or type(node.ops[0]) not in _CMP_OPS:
            raise self._err(node, "Only single binary comparisons are supported")
        return ir.BinaryOp(_CMP_OPS[type(node.ops[0])], self.visit(node.left), self.visit(node.comparators[0]))

    def visit_BoolOp(self, node: ast.BoolOp) -> ir.Expr:
        op = "and" if isinstance(node.op, ast.And) else "or"
        values = [self.visit(v) for v in node.values]
        # right-nested like the reference (gtscript_frontend.py:1558-1571)
        expr = values[-1]
        for v in reversed(values[:-1]):
            expr = ir.BinaryOp(op, v, expr)
        return expr

    def visit_IfExp(self, node: ast.IfExp) -> ir.Expr:
        return ir.TernaryOp(self.visit(node.test), self.visit(node.body), self.visit(node.orelse))

    def _visit_absolute_k(self, node: ast.Call) -> ir.Expr:
        """``field.at(K=<int expression>)``: read at an absolute K level (gtscript_frontend.py:1671-1731)."""
        if not isinstance(node.func.value, ast.Name):
            raise self._err(node, "Absolute K index: Bad syntax. Must be of the form `field.at(...)`")
        if node.args or not node.keywords:
            raise self._err(node, "Absolute K index: Bad syntax. Must be of the form`.at(K=...)`")
        if node.keywords[0].arg != "K":
            raise self._err(node, "Absolute K index: Bad syntax. First argument must be `K`, e.g. `.at(K=...)`.")
        if len(node.keywords) > 2 or (len(node.keywords) == 2 and node.keywords[1].arg != "ddim"):
            raise self._err(node, "Absolute K index: Bad syntax. Second argument (optional) must be `ddim`, "
                                  "e.g. `.at(K=..., ddim=[...])`.")
        if len(node.keywords) == 2 and not isinstance(node.keywords[1].value, ast.List):
            raise self._err(node, "Absolute K index: Bad syntax. Second argument `ddim` (optional) must be a list of "
                                  "values, e.g. `.at(K=..., ddim=[...])`.")
        level = self.visit(node.keywords[0].value)
        if isinstance(level, ir.AxisIndex):
            raise self._err(node, "Absolute K index: bad syntax, you cannot write `.at(K=K)` since `.at` denotes an "
                                  "absolute index, this is equivalent to `field[0, 0, 0]` or simply `field`.")
        name = node.func.value.id
        decl = self.fields.get(name) or self.temporaries.get(name)
        if decl is None:
            raise GTScriptSymbolError(f"Unknown field '{name}' in stencil '{self.definition.__name__}'")
        if "K" not in decl.axes:
            raise ValueError("Tried accessing a field with no K-dimensions with an absolute K-index.")
        data_index: Tuple = ()
        if len(node.keywords) == 2:
            fake = ast.Subscript(value=ast.Attribute(value=node.func.value, attr="A", ctx=ast.Load()),
                                 slice=ast.Tuple(elts=node.keywords[1].value.elts, ctx=ast.Load()), ctx=ast.Load())
            _, data_index = self._split_data_index(ast.copy_location(fake, node))
        elif decl.data_dims:
            raise self._err(node, f"Field {name} has data dimensions but no data dimensions index is specified. "
                                  f"Use `{name}.at(K=..., ddim=[...])`.")
        return ir.FieldAccess(name, (0, 0, 0), None, level, data_index, True)

    def visit_Call(self, node: ast.Call) -> ir.Expr:
        name = self._call_name(node)
        if isinstance(node.func, ast.Attribute) and node.func.attr == "at":
            return self._visit_absolute_k(node)
        if node.keywords:
            raise self._err(node, "Keyword arguments are not supported in calls")
        args = tuple(self.visit(a) for a in node.args)
        if name in _CAST_FUNCS or name in ("int", "float"):
            if len(args) != 1:
                raise self._err(node, f"{name}() takes exactly one argument")
            dt = _CAST_FUNCS.get(name) or (self.int_dtype if name == "int" else self.float_dtype)
            return ir.NativeCall(f"cast:{dt.name}", args, dt)
        if name in _NATIVE_UFUNC:
            return ir.NativeCall(name, args)
        if isinstance(node.func, ast.Name) and callable(self.context.get(name, self.externals.get(name))):
            raise TypeError(f"{name} is not a gtscript function")
        raise self._err(node, f"Unsupported call to '{name}'")


# ---------------------------------------------------------------------------------------------
# typing passes
# ---------------------------------------------------------------------------------------------
def _resolve_and_upcast(stencil: ir.Stencil) -> ir.Stencil:
    """Type every node, then make every dtype transition an explicit Cast."""
    dtypes: Dict[str, Optional[np.dtype]] = {d.name: d.dtype for d in (*stencil.fields, *stencil.temporaries)}
    for p in stencil.params:
        dtypes[p.name] = p.dtype

    def typed(expr: ir.Expr) -> ir.Expr:
        def fn(e: ir.Expr) -> ir.Expr:
            if isinstance(e, ir.FieldAccess):
                dt = dtypes.get(e.name)
                if dt is None:
                    raise GTScriptSymbolError(f"Temporary '{e.name}' is read before it is assigned")
                koff = e.koffset
                if koff is not None:  # already typed (map_expr is bottom-up); must be an integer
                    if np.dtype(koff.dtype).kind not in "iu":
                        raise GTScriptSyntaxError(f"Variable K offset of '{e.name}' must be an integer expression")
                for d in e.data_index or ():
                    if isinstance(d, ir.Expr) and np.dtype(d.dtype).kind not in "iu":
                        raise GTScriptSyntaxError(f"Data index of '{e.name}' must be an integer expression")
                return ir.FieldAccess(e.name, e.offset, dt, koff, e.data_index, e.absolute_k)
            if isinstance(e, (ir.Literal, ir.ScalarAccess, ir.Cast, ir.AxisIndex)):
                return e
            if isinstance(e, ir.UnaryOp):
                ufunc = _OP_UFUNC[{"-": "neg", "+": "pos", "not": "not"}[e.op]]
                (target,) = ufunc_signature(ufunc, (e.expr.dtype,))
                inner = e.expr if e.expr.dtype == target else ir.Cast(e.expr, target)
                return ir.UnaryOp(e.op, inner, np.dtype("bool") if e.op == "not" else target)
            if isinstance(e, ir.BinaryOp) and e.op == "**" and e.left.dtype.kind != e.right.dtype.kind \
                    and "b" not in (e.left.dtype.kind, e.right.dtype.kind):
                # The reference does not upcast the arguments of a power (gtir_upcaster.py:114-126), so a float
                # base with an integer exponent reaches numpy / std::pow as it is; both promote the pair to
                # double (np.result_type(float32, int32) is float64; std::pow(float, int) returns double).
                compute = np.result_type(e.left.dtype, e.right.dtype)
                left = e.left if e.left.dtype == compute else ir.Cast(e.left, compute)
                right = e.right if e.right.dtype == compute else ir.Cast(e.right, compute)
                return ir.BinaryOp("**", left, right, compute)
            if isinstance(e, ir.BinaryOp):
                lt, rt = ufunc_signature(_OP_UFUNC[e.op], (e.left.dtype, e.right.dtype))
                left = e.left if e.left.dtype == lt else ir.Cast(e.left, lt)
                right = e.right if e.right.dtype == rt else ir.Cast(e.right, rt)
                if e.op in ir.ARITHMETIC_OPS:
                    out = _max_dtype(e.left.dtype, e.right.dtype)
                    if e.op == "/" and _RANK[out] < _RANK[np.dtype("float32")]:
                        out = lt  # true division of integers computes (and yields) floats
                else:
                    out = np.dtype("bool")
                return ir.BinaryOp(e.op, left, right, out)
            if isinstance(e, ir.TernaryOp):
                out = _max_dtype(e.true_expr.dtype, e.false_expr.dtype)
                t = e.true_expr if e.true_expr.dtype == out else ir.Cast(e.true_expr, out)
                f = e.false_expr if e.false_expr.dtype == out else ir.Cast(e.false_expr, out)
                return ir.TernaryOp(e.cond, t, f, out)
            if isinstance(e, ir.NativeCall):
                if e.func.startswith("cast:"):
                    return e
                arg_dt = tuple(a.dtype for a in e.args)
                sig = ufunc_signature(_NATIVE_UFUNC[e.func], arg_dt)
                args = tuple(a if a.dtype == s else ir.Cast(a, s) for a, s in zip(e.args, sig))
                out = np.dtype("bool") if e.func in ("isfinite", "isinf", "isnan") else _max_dtype(*sig)
                return ir.NativeCall(e.func, args, out)
            raise TypeError(e)

        return ir.map_expr(expr, fn)

    new_comps = []
    for comp in stencil.computations:
        new_blocks = []
        for block in comp.blocks:
            new_body = []
            for stmt in block.body:
                mask = None
                if stmt.mask is not None:
                    mask = typed(stmt.mask)
                    if mask.dtype != np.dtype("bool"):
                        mask = ir.Cast(mask, np.dtype("bool"))
                value = typed(stmt.value)
                name = stmt.target.name
                if dtypes.get(name) is None:
                    dtypes[name] = value.dtype  # AUTO temporaries: dtype of the first RHS
                tdt = dtypes[name]
                if value.dtype != tdt:
                    value = ir.Cast(value, tdt)
                loops = []
                for lid, cond in stmt.loops:
                    cond = typed(cond)
                    loops.append((lid, cond if cond.dtype == np.dtype("bool") else ir.Cast(cond, np.dtype("bool"))))
                target = typed(stmt.target)
                new_body.append(ir.Assign(target, value, mask, stmt.group, stmt.region, tuple(loops)))
            new_blocks.append(ir.IntervalBlock(block.interval, tuple(new_body)))
        new_comps.append(ir.Computation(comp.order, tuple(new_blocks)))
    temps = tuple(ir.FieldDecl(t.name, dtypes[t.name], t.axes, t.data_dims, False) for t in stencil.temporaries)
    return ir.Stencil(stencil.name, stencil.fields, stencil.params, temps, tuple(new_comps))


def _check_semantics(stencil: ir.Stencil) -> None:
    """The legality checks of the reference's GTIR / OIR validators, with their messages."""
    api = {f.name for f in stencil.fields}
    written = {s.target.name for _, _, s in stencil.statements()}
    horizontal = lambda e: e.offset[0] != 0 or e.offset[1] != 0  # noqa: E731
    for comp, block, stmt in stencil.statements():
        # gtir.py:96-110 (ParAssignStmt)
        if any(isinstance(e, ir.FieldAccess) and e.name == stmt.target.name and horizontal(e) for e in ir.walk(stmt.value)):
            raise ValueError("Self-assignment with offset in I or J is illegal.")
    for comp in stencil.computations:
        for block in comp.blocks:
            # gtir.py:224-241 (VerticalLoop): an API field written and read with a horizontal offset in one loop
            writes = {s.target.name for s in block.body}
            offset_reads = {e.name for s in block.body for e in ir.stmt_reads(s) if isinstance(e, ir.FieldAccess) and horizontal(e)}
            illegal = (writes & offset_reads) & api
            if illegal:
                raise ValueError(f"Illegal write and read with horizontal offset detected for {illegal}.")
            # gtir.py:243-293: K offsets between a write and any other access to the same field in a PARALLEL loop
            iv = block.interval
            if comp.order is not ir.LoopOrder.PARALLEL or (iv.start.level == iv.end.level
                                                           and abs(iv.end.offset - iv.start.offset) == 1):
                continue
            targets = {}
            for s in block.body:
                targets.setdefault(s.target.name, []).append(s.target)
            for s in block.body:
                for e in ir.stmt_reads(s):
                    if not isinstance(e, ir.FieldAccess) or e.name not in targets:
                        continue
                    for w in targets[e.name]:
                        if e.koffset is not None or w.koffset is not None:
                            raise ValueError("Not allowed to write and read with `VariableKOffset` and/or "
                                             f"`AbsoluteKIndex` in PARALLEL loops: `{e.name}`")
                        if e.offset[2] != w.offset[2]:
                            raise ValueError(f"Not allowed to write and read with k-offsets in PARALLEL loops: `{e.name}`")
    for comp, _, stmt in stencil.statements():
        for e in ir.stmt_reads(stmt):
            if not isinstance(e, ir.FieldAccess):
                continue
            # N5: a written API field may not be read with a horizontal offset.  The reference applies this to the
            # read EXTENT (validate_stencil_memory_accesses, gtir_to_oir.py:19-46), which also rejects reads that
            # reach the field through a temporary; its reason is a race in the GridTools backends.  Here stages
            # are cut at exactly those dependencies, so only the direct form is an error: every program the
            # reference accepts is accepted, and the indirect ones run with the numpy backend's semantics.
            if e.name in api and e.name in written and horizontal(e):
                raise ValueError(f"Found non-zero read extent on written fields: {e.name}")


def parse_stencil(definition, *, externals: Dict[str, Any], dtypes: Dict[Any, Any],
                  options: gt_definitions.BuildOptions) -> ir.Stencil:
    """Definition function -> typed, upcast ``ir.Stencil`` (raises on anything outside the subset)."""
    sig = inspect.signature(definition)
    for p in sig.parameters.values():
        if p.kind == inspect.Parameter.VAR_POSITIONAL:
            raise GTScriptDefinitionError("'*args' tuple parameter is not supported in GTScript definitions")
        if p.kind == inspect.Parameter.VAR_KEYWORD:
            raise GTScriptDefinitionError("'**kwargs' dict parameter is not supported in GTScript definitions")
    annotations = gtscript._resolve_annotations(definition, dtypes)
    untyped = _Parser(definition, annotations, externals, options, dtypes).parse()
    typed = _resolve_and_upcast(untyped)
    _check_semantics(typed)
    return typed
