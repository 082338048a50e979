"""ORACLE (test infrastructure, NOT product code): a ``numpy`` backend for gt4py_amd's stencil IR.

Generic restatement of what the reference's numpy backend executes
(/root/reference/src/gt4py/cartesian/gtc/numpy/npir_codegen.py:331-367 module skeleton, :281-318
vertical passes and horizontal blocks, :205-225 assignments / np.where, :243-248 sequential K loops;
/root/reference/src/gt4py/cartesian/utils/field.py:15-74 origin-shifting ``Field`` shim;
/root/reference/src/gt4py/cartesian/gtc/numpy/oir_to_npir.py:42-56 temporaries).

Importing this module REGISTERS backend ``"numpy"`` with ``gt4py_amd.cartesian.backend`` so that the
CPU tests can run the reference's call-interface tests and BASELINE config[0] ("5-point Laplacian via
@gtscript.stencil on backend=numpy") without a GPU.  It interprets the product's parsed IR, so it
shares the product's frontend; the hand-written restatements in ``ref_numpy.py`` are the independent
check (tests/test_oracle.py compares the two).  Nothing under ``gt4py_amd/`` imports this file.
"""

from __future__ import annotations

import inspect
from typing import Any, Dict, Tuple

import numpy as np
import scipy.special

from gt4py_amd.cartesian import analysis, ir
from gt4py_amd.cartesian.backend import base
from gt4py_amd.cartesian.stencil_object import StencilObject
from gt4py_amd.storage import layout as storage_layout

_BIN = {
    "+": np.add, "-": np.subtract, "*": np.multiply, "/": np.true_divide, "%": np.remainder, "**": np.power,
    ">": np.greater, "<": np.less, ">=": np.greater_equal, "<=": np.less_equal, "==": np.equal,
    "!=": np.not_equal, "and": np.logical_and, "or": np.logical_or,
}
_UN = {"-": np.negative, "+": np.positive, "not": np.logical_not}
_NATIVE = {
    "abs": np.abs, "min": np.minimum, "max": np.maximum, "mod": np.remainder, "sin": np.sin, "cos": np.cos,
    "tan": np.tan, "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan, "sinh": np.sinh, "cosh": np.cosh,
    "tanh": np.tanh, "asinh": np.arcsinh, "acosh": np.arccosh, "atanh": np.arctanh, "sqrt": np.sqrt,
    "exp": np.exp, "log": np.log, "log10": np.log10, "cbrt": np.cbrt, "isfinite": np.isfinite,
    "isinf": np.isinf, "isnan": np.isnan, "floor": np.floor, "ceil": np.ceil, "trunc": np.trunc,
    # gtc/ufuncs.py:16-40: scipy.special for erf / erfc / gamma, np.round (ties to even), and
    # copysign(floor(|x| + 0.5), x) for ties away from zero
    "erf": scipy.special.erf, "erfc": scipy.special.erfc, "gamma": scipy.special.gamma, "round": np.round,
    "round_away_from_zero": lambda x: np.copysign(np.floor(np.abs(x) + 0.5), x),
}


class _FieldShim:
    """Array + per-axis origin; missing cartesian axes broadcast (utils/field.py:15-74)."""

    def __init__(self, array: np.ndarray, origin: Tuple[int, ...], axes: Tuple[str, ...]):
        mask = [a in axes for a in "IJK"]
        full_origin, it = [], iter(origin)
        index = []
        for present in mask:
            if present:
                full_origin.append(int(next(it)))
                index.append(slice(None))
            else:
                full_origin.append(0)
                index.append(np.newaxis)
        self.array = array[tuple(index)] if not all(mask) else array
        self.origin = tuple(full_origin)
        self.mask = mask

    def window(self, lo, hi, offset, krange, data_index=()):
        """View over [lo, hi) per horizontal axis (relative to the origin), shifted by offset; ``data_index``
        selects one element of the trailing data dimensions."""
        idx = []
        for ax in range(2):
            if self.mask[ax]:
                a = self.origin[ax] + lo[ax] + offset[ax]
                b = self.origin[ax] + hi[ax] + offset[ax]
                assert a >= 0 and b <= self.array.shape[ax], "numpy oracle: access outside of the array"
                idx.append(slice(a, b))
            else:
                idx.append(slice(None))
        if self.mask[2]:
            a = self.origin[2] + krange[0] + offset[2]
            b = self.origin[2] + krange[1] + offset[2]
            assert a >= 0 and b <= self.array.shape[2], "numpy oracle: K access outside of the array"
            idx.append(slice(a, b))
        else:
            idx.append(slice(None))
        return self.array[tuple(idx) + tuple(data_index or ())]


def _index_arrays(self, lo, hi, offset, krange, kshift=None, absolute=None, data_index=()):
    """Index arrays for (i, j, k', d...) over the window, for numpy advanced indexing: k' = k + kshift[i, j, k]
    (`lk + k`, npir_codegen.py:110, 271-278), or k' = absolute[i, j, k] counted from the field's K origin
    (`field.at(K=...)`); data indices may be arrays too."""
    ni, nj, nk = hi[0] - lo[0], hi[1] - lo[1], krange[1] - krange[0]
    idx = []
    for ax, n in ((0, ni), (1, nj)):
        if self.mask[ax]:
            base = self.origin[ax] + lo[ax] + offset[ax]
            assert base >= 0 and base + n <= self.array.shape[ax], "numpy oracle: access outside of the array"
            shape = [1, 1, 1]
            shape[ax] = n
            idx.append((base + np.arange(n)).reshape(shape))
        else:
            idx.append(np.zeros((1, 1, 1), dtype=np.int64))
    if not self.mask[2]:
        assert absolute is None, "Tried accessing a field with no K-dimensions with an absolute K-index."
        kk = np.zeros((1, 1, 1), dtype=np.int64)
    elif absolute is not None:
        kk = self.origin[2] + np.broadcast_to(np.asarray(absolute), (ni, nj, nk)).astype(np.int64)
    else:
        kk = self.origin[2] + krange[0] + offset[2] + np.arange(nk).reshape(1, 1, nk)
        if kshift is not None:
            kk = kk + np.broadcast_to(np.asarray(kshift), (ni, nj, nk)).astype(np.int64)
    if self.mask[2]:
        assert kk.min() >= 0 and kk.max() < self.array.shape[2], "numpy oracle: K access outside of the array"
    data = []
    for d, n in zip(data_index or (), self.array.shape[3:]):
        d = np.asarray(d).astype(np.int64)
        assert d.min() >= 0 and d.max() < n, "numpy oracle: data index outside of the array"
        data.append(d)
    return (idx[0], idx[1], kk, *data)


_FieldShim.index_arrays = _index_arrays


def _k_index(krange):
    """The K iteration index over the window, counted from the start of the compute domain."""
    return np.arange(krange[0], krange[1]).reshape(1, 1, -1)


def _evaluate(expr: ir.Expr, env, lo, hi, krange):
    def ev(e):
        if isinstance(e, ir.Literal):
            return np.dtype(e.dtype).type(e.value)
        if isinstance(e, ir.FieldAccess):
            return _read(e, env, lo, hi, krange, ev)
        if isinstance(e, ir.AxisIndex):
            return _k_index(krange).astype(e.dtype)
        if isinstance(e, ir.ScalarAccess):
            return env[e.name]
        if isinstance(e, ir.Cast):
            v = ev(e.expr)
            return v.astype(e.dtype) if isinstance(v, np.ndarray) else np.dtype(e.dtype).type(v)
        if isinstance(e, ir.UnaryOp):
            return _UN[e.op](ev(e.expr))
        if isinstance(e, ir.BinaryOp):
            return _BIN[e.op](ev(e.left), ev(e.right))
        if isinstance(e, ir.TernaryOp):
            return np.where(ev(e.cond), ev(e.true_expr), ev(e.false_expr))
        if isinstance(e, ir.NativeCall):
            if e.func.startswith("cast:"):
                v = ev(e.args[0])
                return v.astype(e.dtype) if isinstance(v, np.ndarray) else np.dtype(e.dtype).type(v)
            return _NATIVE[e.func](*[ev(a) for a in e.args])
        raise TypeError(e)

    return ev(expr)


def _needs_index_arrays(e: ir.FieldAccess) -> bool:
    return e.koffset is not None or any(isinstance(d, ir.Expr) for d in e.data_index or ())


def _access_index(e: ir.FieldAccess, env, lo, hi, krange, ev):
    data = tuple(ev(d) if isinstance(d, ir.Expr) else d for d in e.data_index or ())
    k = ev(e.koffset) if e.koffset is not None else None
    return env[e.name].index_arrays(lo, hi, e.offset, krange, None if e.absolute_k else k, k if e.absolute_k else None, data)


def _read(e: ir.FieldAccess, env, lo, hi, krange, ev):
    if _needs_index_arrays(e):
        shim = env[e.name]
        return shim.array[_access_index(e, env, lo, hi, krange, ev)]
    return env[e.name].window(lo, hi, e.offset, krange, e.data_index)


def run_stencil(stencil: ir.Stencil, extents: analysis.ExtentInfo, domain, origin, arrays: Dict[str, Any],
                params: Dict[str, Any]) -> None:
    dI, dJ, dK = (int(d) for d in domain)
    env: Dict[str, Any] = {}
    for decl in stencil.fields:
        if arrays.get(decl.name) is not None:
            env[decl.name] = _FieldShim(arrays[decl.name], tuple(origin[decl.name])[: len(decl.axes)], decl.axes)
    temp_extents = analysis.storage_extents(stencil, extents)
    for decl in stencil.temporaries:
        (ilo, ihi), (jlo, jhi) = temp_extents[decl.name]
        if "K" in decl.axes:
            shape = (dI + (ihi - ilo), dJ + (jhi - jlo), dK, *decl.data_dims)
            env[decl.name] = _FieldShim(np.empty(shape, dtype=decl.dtype), (-ilo, -jlo, 0), ("I", "J", "K"))
        else:  # 2-d temporary: one value per column
            shape = (dI + (ihi - ilo), dJ + (jhi - jlo), *decl.data_dims)
            env[decl.name] = _FieldShim(np.empty(shape, dtype=decl.dtype), (-ilo, -jlo), ("I", "J"))
    for p in stencil.params:
        # a missing (None) parameter stays None: using it raises TypeError inside numpy, exactly what
        # the reference's generated code does when validation was skipped by the call cache
        env[p.name] = None if params.get(p.name) is None else np.dtype(p.dtype).type(params[p.name])

    stmt_blocks = iter(extents.blocks)
    with np.errstate(divide="ignore", over="ignore", under="ignore", invalid="ignore"):
        for comp in stencil.computations:
            for block in comp.blocks:
                k0, k1 = block.interval.range(dK)
                plan = [(stmt, next(stmt_blocks)) for stmt in block.body]

                def box(stmt, block):
                    (ilo, ihi), (jlo, jhi) = block
                    lo, hi = (ilo, jlo), (dI + ihi, dJ + jhi)
                    if stmt.region is not None:
                        # horizontal mask: the block clipped to the region, bounds relative to the
                        # compute domain (horizontal_masks.py:61-112 compute_relative_mask)
                        i0, i1 = stmt.region.i.clip(lo[0], hi[0], dI)
                        j0, j1 = stmt.region.j.clip(lo[1], hi[1], dJ)
                        if i1 <= i0 or j1 <= j0:
                            return None
                        lo, hi = (i0, j0), (i1, j1)
                    return lo, hi

                def run(items, depth, krange, snapshot=None):
                    n = 0
                    while n < len(items):
                        stmt, block = items[n]
                        if len(stmt.loops) > depth:
                            # `while np.any(cond): body` (npir_codegen.py:252-267); consecutive statements
                            # with this loop id are the body
                            lid, cond = stmt.loops[depth]
                            m = n
                            while m < len(items) and len(items[m][0].loops) > depth and items[m][0].loops[depth][0] == lid:
                                m += 1
                            span = box(stmt, block)
                            if span is not None:
                                if lid >= ir.POINTWISE_LOOP:
                                    # the compiled backends' loop: per point, the whole body runs when the condition held at
                                    # the start of the iteration (ir.Assign.loops)
                                    while True:
                                        held = np.array(np.broadcast_to(_evaluate(cond, env, span[0], span[1], krange),
                                                                        (span[1][0] - span[0][0], span[1][1] - span[0][1], krange[1] - krange[0])))
                                        if snapshot is not None:  # a loop inside a per-point loop: only where the outer one runs
                                            outer, (o_lo, o_hi) = snapshot
                                            held &= outer[span[0][0] - o_lo[0]: span[1][0] - o_lo[0], span[0][1] - o_lo[1]: span[1][1] - o_lo[1]]
                                        if not held.any():
                                            break
                                        run(items[n:m], depth + 1, krange, (held, span))
                                else:
                                    while np.any(_evaluate(cond, env, span[0], span[1], krange)):
                                        run(items[n:m], depth + 1, krange, snapshot)
                            n = m
                            continue
                        n += 1
                        span = box(stmt, block)
                        if span is None:
                            continue
                        lo, hi = span
                        scatter = None
                        if _needs_index_arrays(stmt.target):  # run-time K offset / data index of the write
                            ev = lambda e: _evaluate(e, env, lo, hi, krange)  # noqa: E731
                            scatter = _access_index(stmt.target, env, lo, hi, krange, ev)
                            target = env[stmt.target.name].array[scatter]
                        else:
                            target = env[stmt.target.name].window(lo, hi, stmt.target.offset, krange, stmt.target.data_index)
                        held = None
                        if snapshot is not None:  # inside a per-point loop: the part of the snapshot this statement's box covers
                            arr, (s_lo, s_hi) = snapshot
                            held = arr[lo[0] - s_lo[0]: hi[0] - s_lo[0], lo[1] - s_lo[1]: hi[1] - s_lo[1]]
                        if stmt.mask is not None or held is not None:  # npir_codegen.py:205-210: np.where(mask, right, left)
                            mask = _evaluate(stmt.mask, env, lo, hi, krange) if stmt.mask is not None else held
                            if stmt.mask is not None and held is not None:
                                mask = np.logical_and(held, mask)
                            value = np.where(mask, _evaluate(stmt.value, env, lo, hi, krange), target)
                        else:
                            value = _evaluate(stmt.value, env, lo, hi, krange)
                        if scatter is not None:
                            env[stmt.target.name].array[scatter] = value
                        else:
                            target[...] = value

                def execute(krange):
                    run(plan, 0, krange)

                if comp.order is ir.LoopOrder.PARALLEL:
                    if k1 > k0:
                        execute((k0, k1))
                elif comp.order is ir.LoopOrder.FORWARD:
                    for k in range(k0, k1):
                        execute((k, k + 1))
                else:
                    for k in range(k1 - 1, k0 - 1, -1):
                        execute((k, k + 1))


class NumpyOracleStencilObject(StencilObject):
    def _run_implementation(self, domain, origin, exec_info, arguments):
        cls = type(self)
        arrays = {n: arguments.get(n) for n in cls._gt_field_info_}
        params = {n: arguments.get(n) for n in cls._gt_parameter_info_}
        run_stencil(cls._oracle_ir_, cls._oracle_extents_, domain, origin, arrays, params)


class NumpyOracleBackend(base.BaseBackend):
    name = "numpy"
    options = {"ignore_np_errstate": {"versioning": True, "type": bool}, "while_loops": {"versioning": True, "type": str}}
    storage_info = storage_layout.from_name("numpy")
    languages = {"computation": "python", "bindings": ["python"]}

    def make_stencil_class(self):
        b = self.builder
        sig = inspect.signature(b.definition)
        sig = sig.replace(parameters=[p.replace(annotation=inspect.Parameter.empty) for p in sig.parameters.values()])
        attrs = {
            "_gt_id_": b.stencil_id,
            "definition_func": staticmethod(b.definition),
            "_gt_backend_": self.name,
            "_gt_source_": b.source,
            "_gt_domain_info_": b.args_data.domain_info,
            "_gt_field_info_": b.args_data.field_info,
            "_gt_parameter_info_": b.args_data.parameter_info,
            "_gt_constants_": dict(b.externals),
            "_gt_options_": b.options.as_dict(),
            "_gt_signature_": sig,
            "_oracle_ir_": b.stencil_ir,
            "_oracle_extents_": b.args_data.extents,
            "__module__": b.options.module,
        }
        return type(b.class_name, (NumpyOracleStencilObject,), attrs)


def register() -> None:
    """Register the oracle as backend "numpy" (idempotent)."""
    if "numpy" not in base.REGISTRY:
        base.register(NumpyOracleBackend)


register()
