"""AST-level inlining of ``@gtscript.function`` calls, done before the stencil body is parsed.

Follows the reference's ``CallInliner``
(/root/reference/src/gt4py/cartesian/frontend/gtscript_frontend.py:505-757):

* the callee's statements are spliced into the caller's block *in front of* the statement holding the
  call;
* a parameter bound to a plain name that the callee never assigns is renamed to the caller's name
  (:658-662); any other argument (subscripts like ``u[1, 0, 0]``, expressions, constants) is first
  assigned to a fresh name (:699-717), so offsets inside the callee apply to that temporary;
* names assigned inside the callee are made unique per call site (:664-673);
* ``return`` becomes an assignment to the call's target, or to a fresh ``RETURN_VALUE`` name when the
  call sits inside an expression (:676-697); only single-valued functions may be used that way;
* recursion is an error (:596-600); defaults and keyword arguments are honoured (:619-641).

Only values matter here (extents and temporaries are recomputed from the flattened body), so the one
liberty taken is in multi-value returns: each returned expression is first stored in its own fresh
name and then copied to the targets, which keeps ``a, b = f(b, a)``-style calls simultaneous.
"""

from __future__ import annotations

import ast
import copy
import inspect
import textwrap
from typing import Any, Dict, List, Optional, Set

import numbers

import numpy as np

from .definitions import GTScriptDefinitionError, GTScriptSyntaxError

#: names that mean something to GTScript itself and are never looked up in the enclosing Python scope
#: (the reference's ``gtscript.builtins``, gtscript.py:78-103)
GTSCRIPT_BUILTINS = frozenset({
    "I", "J", "K", "IJ", "IK", "JK", "IJK", "FORWARD", "BACKWARD", "PARALLEL", "Field", "GlobalTable", "Sequence",
    "externals", "computation", "interval", "horizontal", "region", "__gtscript__", "__externals__", "__INLINED",
    "compile_assert", "abs", "min", "max", "mod", "sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh",
    "asinh", "acosh", "atanh", "sqrt", "exp", "log", "log10", "gamma", "cbrt", "isfinite", "isinf", "isnan", "floor",
    "ceil", "trunc", "erf", "erfc", "round", "round_away_from_zero", "int32", "int64", "float32", "float64", "int",
    "float", "bool", "True", "False", "None",
})


def is_gtscript_function(obj: Any) -> bool:
    return callable(obj) and getattr(obj, "__gtscript_function__", False)


def _unlazy(obj: Any) -> Any:
    """``@gtscript.lazy_function()`` hands out a zero-argument callable that annotates and returns the function the
    first time it is needed (gtscript_frontend.py:2349-2357)."""
    if callable(obj) and not is_gtscript_function(obj) and getattr(obj, "__qualname__", "").startswith("lazy_function."):
        return obj()
    return obj


def _is_constant(value: Any) -> bool:
    return isinstance(value, (bool, np.bool_, numbers.Real)) or (isinstance(value, np.generic) and value.dtype.kind in "biuf")


def _dotted(node: ast.AST) -> Optional[str]:
    if isinstance(node, ast.Name):
        return node.id
    if isinstance(node, ast.Attribute):
        base = _dotted(node.value)
        return f"{base}.{node.attr}" if base else None
    return None


class _NonlocalBinder(ast.NodeTransformer):
    """Replace names from the enclosing Python scope by their VALUES: numeric constants (``GRAV``, ``consts.A``,
    ``Config.nested.B`` ...) become literals that keep their numpy type; gtscript functions stay names for the call
    inliner; anything else is an error (GTScriptParser.collect_external_symbols / eval_external,
    gtscript_frontend.py:2269-2375)."""

    def __init__(self, context: Dict[str, Any], known: Set[str]):
        self.context = context
        self.known = known  # parameters, assigned names, names imported from __externals__

    def _bind(self, node: ast.AST):
        name = _dotted(node)
        if name is None:
            return self.generic_visit(node)
        root = name.split(".")[0]
        if root in self.known or root in GTSCRIPT_BUILTINS or root not in self.context:
            return node
        try:
            value = self.context[root]
            for attr in name.split(".")[1:]:
                value = getattr(value, attr)
        except AttributeError as ex:
            raise GTScriptDefinitionError(f"Missing or invalid value for external symbol {name}") from ex
        from . import gtscript

        value = _unlazy(value)
        if is_gtscript_function(value) or isinstance(value, gtscript.Axis) or (
                isinstance(value, type) and value in gtscript.ENUM_REGISTER.values()):
            return node
        if isinstance(node, ast.Attribute) and isinstance(value, type) and root in gtscript.ENUM_REGISTER:
            return node
        if _is_constant(value):
            return ast.copy_location(ast.Constant(value=value), node)
        if value is None or isinstance(value, (type, type(np))) or callable(value):
            return node  # classes / modules / plain callables: only legal as the root of something else
        raise GTScriptDefinitionError(f"Missing or invalid value for external symbol {name} (a {type(value).__name__})")

    def visit_Name(self, node: ast.Name):
        return self._bind(node) if isinstance(node.ctx, ast.Load) else node

    def visit_Attribute(self, node: ast.Attribute):
        return self._bind(node) if isinstance(node.ctx, ast.Load) else node

    def visit_AnnAssign(self, node: ast.AnnAssign):
        if node.value is not None:
            node.value = self.visit(node.value)
        return node  # the annotation is evaluated as Python by the parser

    def visit_ImportFrom(self, node):
        return node


def _assigned_names(fdef: ast.AST) -> Set[str]:
    out: Set[str] = set()
    for n in ast.walk(fdef):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            out.add(n.id)
        elif isinstance(n, ast.ImportFrom):
            out.update(a.asname or a.name for a in n.names)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
    return out


def bind_nonlocals(fdef: ast.FunctionDef, context: Dict[str, Any]) -> None:
    binder = _NonlocalBinder(context, _assigned_names(fdef))
    fdef.body = [binder.visit(s) for s in fdef.body]


def _function_ast(func) -> ast.FunctionDef:
    tree = ast.parse(textwrap.dedent(inspect.getsource(func)))
    fdef = next(n for n in tree.body if isinstance(n, ast.FunctionDef))
    fdef.decorator_list = []
    return fdef


def _context_of(func) -> Dict[str, Any]:
    frozen = getattr(func, "__gtscript_context__", None)
    if frozen is not None:  # what the names meant when `@gtscript.function` was applied
        return dict(frozen)
    ctx = dict(getattr(func, "__globals__", {}))
    closure = getattr(func, "__closure__", None)
    if closure:
        for name, cell in zip(func.__code__.co_freevars, closure):
            try:
                ctx[name] = cell.cell_contents
            except ValueError:  # empty cell
                pass
    return ctx


class _Renamer(ast.NodeTransformer):
    def __init__(self, mapping: Dict[str, str]):
        self.mapping = mapping

    def visit_Name(self, node: ast.Name):
        if node.id in self.mapping:
            return ast.copy_location(ast.Name(id=self.mapping[node.id], ctx=node.ctx), node)
        return node


class CallInliner:
    """Rewrites a stencil's FunctionDef in place; ``imports`` collects ``from __externals__`` nodes of callees."""

    def __init__(self, context: Dict[str, Any], function_externals: Optional[Dict[str, Any]] = None):
        self.context = context
        self.function_externals = dict(function_externals or {})  # gtscript functions passed as externals
        self.imports: List[ast.ImportFrom] = []
        self._counter = 0

    # -- resolution ---------------------------------------------------------------------------
    @staticmethod
    def _resolve(node: ast.AST, context: Dict[str, Any]) -> Optional[Any]:
        if isinstance(node, ast.Name):
            return _unlazy(context.get(node.id))
        if isinstance(node, ast.Attribute):
            base = CallInliner._resolve(node.value, context)
            return _unlazy(getattr(base, node.attr, None)) if base is not None else None
        return None

    # -- statements ---------------------------------------------------------------------------
    def process_function(self, fdef: ast.FunctionDef) -> ast.FunctionDef:
        fdef.body = self._process_stmts(fdef.body, self.context, frozenset())
        return fdef

    def _process_stmts(self, stmts: List[ast.stmt], context, stack) -> List[ast.stmt]:
        out: List[ast.stmt] = []
        for s in stmts:
            if isinstance(s, (ast.With, ast.If, ast.While)):
                if isinstance(s, ast.If):
                    if any(isinstance(n, ast.Call) and is_gtscript_function(self._resolve(n.func, context))
                           for n in ast.walk(s.test)):
                        # gtscript_frontend.py:552-566 (CallInliner.visit_If)
                        raise GTScriptSyntaxError("Using function calls in the condition of an if is not allowed")
                s.body = self._process_stmts(s.body, context, stack)
                if getattr(s, "orelse", None):
                    s.orelse = self._process_stmts(s.orelse, context, stack)
                out.append(s)
            elif isinstance(s, ast.Assign) and isinstance(s.value, ast.Call) and \
                    is_gtscript_function(self._resolve(s.value.func, context)):
                if len(s.targets) != 1:
                    raise GTScriptSyntaxError("Assignment to more than one target is not supported.")
                self._inline(s.value, out, context, stack, target=s.targets[0])
            elif isinstance(s, (ast.Assign, ast.AugAssign)):
                s.value = self._expr(s.value, out, context, stack)
                out.append(s)
            elif isinstance(s, ast.Return):
                if s.value is not None:
                    s.value = self._expr(s.value, out, context, stack)
                out.append(s)
            else:
                out.append(s)
        return out

    # -- expressions --------------------------------------------------------------------------
    def _expr(self, node: ast.expr, block: List[ast.stmt], context, stack) -> ast.expr:
        inliner = self

        class T(ast.NodeTransformer):
            def visit_Call(self, call: ast.Call):
                call.args = [self.visit(a) for a in call.args]
                for kw in call.keywords:
                    kw.value = self.visit(kw.value)
                if is_gtscript_function(inliner._resolve(call.func, context)):
                    return inliner._inline(call, block, context, stack, target=None)
                return call

        return T().visit(node)

    # -- the inlining itself ------------------------------------------------------------------
    def _inline(self, call: ast.Call, block: List[ast.stmt], context, stack, target: Optional[ast.expr]) -> ast.expr:
        func = self._resolve(call.func, context)
        name = getattr(func, "__name__", "function")
        if func in stack:
            raise GTScriptSyntaxError(f"Found recursive function call '{name}' in the stack.")
        # arguments may themselves hold calls
        call.args = [self._expr(a, block, context, stack) for a in call.args]
        for kw in call.keywords:
            kw.value = self._expr(kw.value, block, context, stack)

        fdef = copy.deepcopy(_function_ast(func))
        callee_context = _context_of(func)
        callee_context.update({k: v for k, v in self.function_externals.items()})
        bind_nonlocals(fdef, callee_context)
        params = [a.arg for a in fdef.args.args] + [a.arg for a in fdef.args.kwonlyargs]
        positional = [a.arg for a in fdef.args.args]
        defaults: Dict[str, ast.expr] = {}
        for arg, d in zip(reversed(fdef.args.args), reversed(fdef.args.defaults)):
            defaults[arg.arg] = d
        for arg, d in zip(fdef.args.kwonlyargs, fdef.args.kw_defaults):
            if d is not None:
                defaults[arg.arg] = d
        bound: Dict[str, ast.expr] = {}
        if len(call.args) > len(positional):
            raise GTScriptSyntaxError(f"Invalid call signature when calling {name}")
        for pname, value in zip(positional, call.args):
            bound[pname] = value
        for kw in call.keywords:
            if kw.arg not in params or kw.arg in bound:
                raise GTScriptSyntaxError(f"Invalid call signature when calling {name}")
            bound[kw.arg] = kw.value
        for pname in params:
            if pname not in bound:
                if pname not in defaults:
                    raise GTScriptSyntaxError(f"Invalid call signature when calling {name}")
                bound[pname] = copy.deepcopy(defaults[pname])

        assigned: Set[str] = set()
        for n in ast.walk(fdef):
            targets = n.targets if isinstance(n, ast.Assign) else [n.target] if isinstance(n, ast.AugAssign) else []
            for t in targets:
                for leaf in ([t] if not isinstance(t, ast.Tuple) else t.elts):
                    if isinstance(leaf, ast.Name):
                        assigned.add(leaf.id)
                    elif isinstance(leaf, ast.Subscript) and isinstance(leaf.value, ast.Name):
                        assigned.add(leaf.value.id)
                    else:
                        raise GTScriptSyntaxError("Unsupported assignment target.")

        self._counter += 1
        suffix = f"__{name}_{getattr(call, 'lineno', 0)}_{getattr(call, 'col_offset', 0)}_{self._counter}"
        mapping: Dict[str, str] = {}
        pre: List[ast.stmt] = []
        for pname, value in bound.items():
            if isinstance(value, ast.Name) and pname not in assigned:
                mapping[pname] = value.id
            else:
                mapping[pname] = pname + suffix
                pre.append(ast.copy_location(
                    ast.Assign(targets=[ast.Name(id=mapping[pname], ctx=ast.Store())], value=value, lineno=call.lineno), call))
        for local in assigned:
            mapping.setdefault(local, local + suffix)

        body = [s for s in fdef.body
                if not (isinstance(s, ast.Expr) and isinstance(s.value, ast.Constant) and isinstance(s.value.value, str))]
        kept: List[ast.stmt] = []
        for s in body:
            if isinstance(s, ast.ImportFrom):
                self.imports.append(s)  # `from __externals__ import X` of the callee: same externals dict
            else:
                kept.append(_Renamer(mapping).visit(s))
        # nested calls resolve in the CALLEE's namespace
        kept = self._process_stmts(kept, callee_context, stack | {func})

        n_returns = sum(isinstance(n, ast.Return) for s in kept for n in ast.walk(s))
        if n_returns != 1 or not isinstance(kept[-1], ast.Return) or kept[-1].value is None:
            raise GTScriptSyntaxError(f"Each gtscript function should have a single return statement as its last "
                                      f"statement ('{name}' has {n_returns})")
        ret = kept.pop().value
        block.extend(pre)
        block.extend(kept)

        def store(node_name: str) -> ast.Name:
            return ast.copy_location(ast.Name(id=node_name, ctx=ast.Store()), call)

        def load(node_name: str) -> ast.Name:
            return ast.copy_location(ast.Name(id=node_name, ctx=ast.Load()), call)

        if isinstance(ret, ast.Tuple):
            if target is None:
                raise GTScriptSyntaxError(
                    "Only functions with a single return value can be used in expressions, including as call "
                    "arguments. Please assign the function results to symbols first.")
            if not isinstance(target, ast.Tuple) or len(target.elts) != len(ret.elts):
                raise GTScriptSyntaxError(f"Number of returns values does not match arguments on left side "
                                          f"('{name}' returns {len(ret.elts)})")
            tmp_names = [f"RETURN_VALUE_{n}{suffix}" for n in range(len(ret.elts))]
            for tmp, value in zip(tmp_names, ret.elts):
                block.append(ast.copy_location(ast.Assign(targets=[store(tmp)], value=value, lineno=call.lineno), call))
            for t, tmp in zip(target.elts, tmp_names):
                block.append(ast.copy_location(ast.Assign(targets=[t], value=load(tmp), lineno=call.lineno), call))
            return load(tmp_names[0])
        if target is not None:
            if isinstance(target, ast.Tuple):
                raise GTScriptSyntaxError(f"Number of returns values does not match arguments on left side "
                                          f"('{name}' returns one value)")
            block.append(ast.copy_location(ast.Assign(targets=[target], value=ret, lineno=call.lineno), call))
            return ret
        tmp = f"RETURN_VALUE{suffix}"
        block.append(ast.copy_location(ast.Assign(targets=[store(tmp)], value=ret, lineno=call.lineno), call))
        return load(tmp)


def inline_calls(fdef: ast.FunctionDef, definition, externals: Optional[Dict[str, Any]] = None) -> List[ast.ImportFrom]:
    """Inline every gtscript-function call under ``fdef`` (the stencil's own AST).  Functions may also
    arrive as externals (test_suites.py:264-285).  Returns the callees' ``from __externals__ import ...``
    nodes so the parser can bind them."""
    context = _context_of(definition)
    functions = {k: _unlazy(v) for k, v in (externals or {}).items() if is_gtscript_function(_unlazy(v))}
    context.update(functions)
    bind_nonlocals(fdef, context)
    inliner = CallInliner(context, functions)
    inliner.process_function(fdef)
    ast.fix_missing_locations(fdef)
    return inliner.imports
