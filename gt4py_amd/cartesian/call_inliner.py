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

from .definitions import GTScriptSyntaxError


def is_gtscript_function(obj: Any) -> bool:
    return callable(obj) and getattr(obj, "__gtscript_function__", False)


def _function_ast(func) -> ast.FunctionDef:
    tree = ast.parse(textwrap.dedent(inspect.getsource(func)))
    fdef = next(n for n in tree.body if isinstance(n, ast.FunctionDef))
    fdef.decorator_list = []
    return fdef


def _context_of(func) -> Dict[str, Any]:
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

    def __init__(self, context: Dict[str, Any]):
        self.context = context
        self.imports: List[ast.ImportFrom] = []
        self._counter = 0

    # -- resolution ---------------------------------------------------------------------------
    @staticmethod
    def _resolve(node: ast.AST, context: Dict[str, Any]) -> Optional[Any]:
        if isinstance(node, ast.Name):
            return context.get(node.id)
        if isinstance(node, ast.Attribute):
            base = CallInliner._resolve(node.value, context)
            return getattr(base, node.attr, None) if base is not None else None
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
                    pre: List[ast.stmt] = []
                    s.test = self._expr(s.test, pre, context, stack)
                    out.extend(pre)
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
        kept = self._process_stmts(kept, _context_of(func), stack | {func})

        if not kept or not isinstance(kept[-1], ast.Return) or kept[-1].value is None:
            raise GTScriptSyntaxError(f"gtscript function '{name}' must end with a 'return' of its value(s)")
        if any(isinstance(n, ast.Return) for s in kept[:-1] for n in ast.walk(s)):
            raise GTScriptSyntaxError(f"gtscript function '{name}': only a single trailing 'return' is supported")
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
                raise GTScriptSyntaxError(f"'{name}' returns {len(ret.elts)} values; the assignment target does not match")
            tmp_names = [f"RETURN_VALUE_{n}{suffix}" for n in range(len(ret.elts))]
            for tmp, value in zip(tmp_names, ret.elts):
                block.append(ast.copy_location(ast.Assign(targets=[store(tmp)], value=value, lineno=call.lineno), call))
            for t, tmp in zip(target.elts, tmp_names):
                block.append(ast.copy_location(ast.Assign(targets=[t], value=load(tmp), lineno=call.lineno), call))
            return load(tmp_names[0])
        if target is not None:
            if isinstance(target, ast.Tuple):
                raise GTScriptSyntaxError(f"'{name}' returns one value; the assignment target is a tuple")
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
    context.update({k: v for k, v in (externals or {}).items() if is_gtscript_function(v)})
    inliner = CallInliner(context)
    inliner.process_function(fdef)
    ast.fix_missing_locations(fdef)
    return inliner.imports
