"""Bridge to a REAL gt4py install: run stencils that gt4py itself parsed and lowered on ``hip:mi300``.

SURVEY.md section 8(f) rank 3.  A gt4py user keeps ``from gt4py.cartesian import gtscript`` -- gt4py's own
frontend, GTIR passes and ``StencilObject`` -- and only the backend changes: ``register_with_gt4py()`` adds a
``BaseBackend`` subclass named ``"hip:mi300"`` to gt4py's registry
(/root/reference/src/gt4py/cartesian/backend/base.py:142-227).  Its ``generate()`` takes the stencil's OIR (what
every gtc backend starts from, e.g. backend/numpy_backend.py:62-77), translates it with ``oir_to_ir`` into this
repo's IR, and hands that to the same planner / code generator / kernel library as the repo's own frontend.

What can be verified where:

* ``oir_to_ir`` only looks at node CLASS NAMES and ATTRIBUTES (gtc/oir.py:33-360, gtc/common.py:65-890), never
  imports gt4py, and is unit-tested here on hand-built OIR-shaped trees (tests/test_adapter.py) -- the translated
  IR must equal what this repo's frontend produces for the same stencil, and runs on the oracle and on the GPU.
* ``register_with_gt4py`` needs gt4py (Python >= 3.12 and its dependencies, absent from this image): it raises
  ImportError with that explanation otherwise.  INTEGRATION.md shows the three lines a user adds.  What is pinned without
  running it: every gt4py name it and ``_wrap_for_gt4py`` touch -- module attributes, ``StencilBuilder`` properties,
  ``BuildOptions`` / ``StencilID`` / ``ModuleData`` fields, the keyword arguments of ``StencilObject._call_run``, the members a
  ``StencilObject`` subclass must define -- is checked against ``tests/golden/gt4py_api_surface.json``, which
  ``scripts/make_gt4py_api_surface.py`` writes from the reference's sources with ``ast`` (tests/test_adapter.py; the check
  found ``builder.backend_name``, which the reference does not have: it is ``builder.backend.name``).  And they EXECUTE, against a
  double that offers exactly the names of that surface (tests/gt4py_double.py): registration, ``generate()``, the generated
  ``StencilObject`` subclass, and on the device a call through gt4py's ``_call_run`` protocol down to the kernel.
"""

from __future__ import annotations

import inspect
import itertools
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from ..cartesian import analysis, definitions as gt_definitions, ir

_DTYPES = {
    "BOOL": np.dtype("bool"), "INT8": np.dtype("int8"), "INT16": np.dtype("int16"), "INT32": np.dtype("int32"),
    "INT64": np.dtype("int64"), "FLOAT32": np.dtype("float32"), "FLOAT64": np.dtype("float64"),
}
#: gtc.common.NativeFunction values -> this repo's names (gtc/common.py:150-190)
_NATIVE = {
    "arcsin": "asin", "arccos": "acos", "arctan": "atan", "arcsinh": "asinh", "arccosh": "acosh", "arctanh": "atanh",
}
_CASTS = {"int32", "int64", "float32", "float64"}


class UnsupportedOIR(NotImplementedError):
    pass


def _kind(node: Any) -> str:
    return type(node).__name__


def _enum_name(value: Any) -> str:
    """'FLOAT64' from DataType.FLOAT64, an int-valued enum, or a plain string."""
    return getattr(value, "name", None) or str(value)


def _enum_value(value: Any) -> str:
    """'forward' from LoopOrder.FORWARD (StrEnum) or a plain string."""
    return str(getattr(value, "value", value))


def _dtype(value: Any) -> np.dtype:
    try:
        return _DTYPES[_enum_name(value).upper()]
    except KeyError:
        raise UnsupportedOIR(f"data type {value!r}") from None


def _axes(dimensions: Sequence[bool]) -> Tuple[str, ...]:
    return tuple(a for a, present in zip("IJK", dimensions) if present)


class _Translator:
    def __init__(self, stencil: Any):
        self.stencil = stencil
        self.loop_ids = itertools.count()
        self.scalars: Dict[str, np.dtype] = {}  # LocalScalar declarations -> thread-local temporaries

    # -- expressions ------------------------------------------------------------------------------------
    def expr(self, e: Any) -> ir.Expr:
        kind = _kind(e)
        if kind == "Literal":
            dt = _dtype(e.dtype)
            raw = _enum_value(e.value)
            if dt == np.dtype("bool"):
                return ir.Literal(raw.lower() in ("true", "1"), dt)
            return ir.Literal(dt.type(raw).item() if dt.kind == "f" else int(float(raw)), dt)
        if kind == "FieldAccess":
            return self.access(e)
        if kind == "ScalarAccess":
            if str(e.name) in self.scalars:  # a LocalScalar: one value per point, like a temporary
                return ir.FieldAccess(str(e.name), (0, 0, 0), self.scalars[str(e.name)])
            return ir.ScalarAccess(str(e.name), _dtype(e.dtype))
        if kind == "IteratorAccess":
            if _enum_value(e.name) != "K":
                raise UnsupportedOIR("iterator access on a parallel axis")
            return ir.AxisIndex("K", _dtype(e.dtype))
        if kind == "UnaryOp":
            op = _enum_value(e.op)
            return ir.UnaryOp(op, self.expr(e.expr), np.dtype("bool") if op == "not" else _dtype(e.dtype))
        if kind == "BinaryOp":
            return ir.BinaryOp(_enum_value(e.op), self.expr(e.left), self.expr(e.right), _dtype(e.dtype))
        if kind == "TernaryOp":
            return ir.TernaryOp(self.expr(e.cond), self.expr(e.true_expr), self.expr(e.false_expr), _dtype(e.dtype))
        if kind == "Cast":
            return ir.Cast(self.expr(e.expr), _dtype(e.dtype))
        if kind == "NativeFuncCall":
            func = _enum_value(e.func)
            args = tuple(self.expr(a) for a in e.args)
            if func in _CASTS:
                return ir.NativeCall(f"cast:{func}", args, np.dtype(func))
            if func == "pow":
                return ir.BinaryOp("**", args[0], args[1], _dtype(e.dtype))
            return ir.NativeCall(_NATIVE.get(func, func), args, _dtype(e.dtype))
        raise UnsupportedOIR(f"expression node {kind}")

    def access(self, e: Any) -> ir.FieldAccess:
        off = e.offset
        data_index = tuple(int(_enum_value(d.value)) if _kind(d) == "Literal" else self.expr(d) for d in (e.data_index or ()))
        dt = _dtype(e.dtype)
        kind = _kind(off)
        if kind == "CartesianOffset":
            return ir.FieldAccess(str(e.name), (int(off.i), int(off.j), int(off.k)), dt, None, data_index)
        if kind == "VariableKOffset":
            return ir.FieldAccess(str(e.name), (0, 0, 0), dt, self.expr(off.k), data_index)
        if kind == "AbsoluteKIndex":
            level = off.k if not isinstance(off.k, int) else None
            k = ir.Literal(int(off.k), np.dtype("int64")) if level is None else self.expr(level)
            return ir.FieldAccess(str(e.name), (0, 0, 0), dt, k, data_index, True)
        raise UnsupportedOIR(f"offset node {kind}")

    # -- statements -------------------------------------------------------------------------------------
    @staticmethod
    def bound(b: Any) -> ir.AxisBound:
        if _kind(b) != "AxisBound":
            raise UnsupportedOIR("run-time interval bounds")
        return ir.AxisBound(ir.Level.START if _enum_value(b.level) == "start" else ir.Level.END, int(b.offset))

    def horizontal_interval(self, iv: Any) -> ir.HorizontalInterval:
        return ir.HorizontalInterval(None if iv.start is None else self.bound(iv.start),
                                     None if iv.end is None else self.bound(iv.end))

    def statements(self, body: Sequence[Any], group: int, mask: Optional[ir.Expr], region: Optional[ir.Region],
                   loops: Tuple[Tuple[int, ir.Expr], ...]) -> List[ir.Assign]:
        out: List[ir.Assign] = []
        for stmt in body:
            kind = _kind(stmt)
            if kind == "AssignStmt":
                left = stmt.left
                if _kind(left) == "ScalarAccess":
                    target = ir.FieldAccess(str(left.name), (0, 0, 0), self.scalars[str(left.name)])
                else:
                    target = self.access(left)
                out.append(ir.Assign(target, self.expr(stmt.right), mask, group, region, loops))
            elif kind == "MaskStmt":
                cond = self.expr(stmt.mask)
                out.extend(self.statements(stmt.body, group, cond if mask is None else ir.BinaryOp("and", mask, cond, np.dtype("bool")),
                                           region, loops))
            elif kind == "While":
                cond = self.expr(stmt.cond)
                full = cond if mask is None else ir.BinaryOp("and", mask, cond, np.dtype("bool"))
                out.extend(self.statements(stmt.body, group, full, region, loops + ((next(self.loop_ids), full),)))
            elif kind == "HorizontalRestriction":
                if region is not None:
                    raise UnsupportedOIR("nested horizontal restrictions")
                r = ir.Region(self.horizontal_interval(stmt.mask.i), self.horizontal_interval(stmt.mask.j))
                out.extend(self.statements(stmt.body, group, mask, r, loops))
            else:
                raise UnsupportedOIR(f"statement node {kind}")
        return out

    def run(self) -> Tuple[ir.Stencil, Tuple[str, ...]]:
        st = self.stencil
        fields, params, order = [], [], []
        for decl in st.params:
            order.append(str(decl.name))
            if _kind(decl) == "FieldDecl":
                fields.append(ir.FieldDecl(str(decl.name), _dtype(decl.dtype), _axes(decl.dimensions),
                                           tuple(int(n) for n in decl.data_dims), True))
            else:
                params.append(ir.ScalarDecl(str(decl.name), _dtype(decl.dtype)))
        temporaries = [ir.FieldDecl(str(t.name), _dtype(t.dtype), _axes(t.dimensions), tuple(int(n) for n in t.data_dims), False)
                       for t in st.declarations]
        groups = itertools.count()
        computations: List[ir.Computation] = []
        for loop in st.vertical_loops:
            order_name = _enum_value(loop.loop_order)
            loop_order = {"parallel": ir.LoopOrder.PARALLEL, "forward": ir.LoopOrder.FORWARD,
                          "backward": ir.LoopOrder.BACKWARD}[order_name]
            blocks = []
            for section in loop.sections:
                body: List[ir.Assign] = []
                for hexec in section.horizontal_executions:
                    for local in hexec.declarations:
                        name = str(local.name)
                        if name not in self.scalars:
                            self.scalars[name] = _dtype(local.dtype)
                            temporaries.append(ir.FieldDecl(name, _dtype(local.dtype), ("I", "J", "K"), (), False))
                    # one horizontal execution = one compute extent = one `group`; a lone assignment is its own
                    # execution, which the frontend of this repo spells -1
                    lone = len(hexec.body) == 1 and _kind(hexec.body[0]) == "AssignStmt"
                    body.extend(self.statements(hexec.body, -1 if lone else next(groups), None, None, ()))
                blocks.append(ir.IntervalBlock(ir.Interval(self.bound(section.interval.start), self.bound(section.interval.end)),
                                               tuple(body)))
            computations.append(ir.Computation(loop_order, tuple(blocks)))
        stencil = ir.Stencil(str(st.name), tuple(fields), tuple(params), tuple(temporaries), tuple(computations))
        return stencil, tuple(order)


def oir_to_ir(oir_stencil: Any) -> Tuple[ir.Stencil, Tuple[str, ...]]:
    """gt4py ``oir.Stencil`` (or anything shaped like one) -> (this repo's typed ``ir.Stencil``, argument order).

    OIR is already typed and upcast (every dtype transition is an explicit Cast, gtir_upcaster.py), its run-time
    ``if`` statements are mask temporaries + ``MaskStmt`` bodies (gtir_to_oir.py:146-232), and one
    ``HorizontalExecution`` is one compute extent: exactly the invariants of ``ir.Stencil`` after this repo's own
    frontend, so the translation is structural."""
    return _Translator(oir_stencil).run()


def stencil_class_from_ir(stencil: ir.Stencil, order: Sequence[str], *, backend: str = "hip:mi300",
                          name: Optional[str] = None, backend_opts: Optional[Dict[str, Any]] = None,
                          constants: Optional[Dict[str, Any]] = None):
    """Build a stencil class for ``backend`` straight from an IR (no GTScript source involved)."""
    from ..cartesian import backend as gt_backend, loader

    name = name or stencil.name
    options = gt_definitions.BuildOptions(name=name, module=__name__, backend_opts=dict(backend_opts or {}))
    # a stand-in with the right call signature: StencilObject.__call__ binds arguments through it
    keyword_only = {p.name for p in stencil.params}
    positional = [n for n in order if n not in keyword_only] + [n for n in order if n in keyword_only]
    namespace: Dict[str, Any] = {}
    exec(f"def {name}({', '.join(positional)}):\n    pass\n", namespace)  # noqa: S102 - identifiers from the IR only
    definition = namespace[name]
    args_data = analysis.make_args_data(stencil)
    source = "\n".join(f"{s.target.name}[...] = {ir.fmt(s.value)}" for _, _, s in stencil.statements())
    import hashlib

    stencil_id = hashlib.sha256((repr(stencil) + backend + repr(sorted((backend_opts or {}).items()))).encode()).hexdigest()
    builder = loader.StencilBuilder(definition, backend, options, dict(constants or {}), {}, stencil, args_data, stencil_id, source)
    return gt_backend.from_name(backend)(builder).generate()


def stencil_from_oir(oir_stencil: Any, **kwargs):
    """OIR -> ready-to-call stencil object on ``hip:mi300`` (or the backend given by ``backend=``)."""
    stencil, order = oir_to_ir(oir_stencil)
    return stencil_class_from_ir(stencil, order, **kwargs)()


def register_with_gt4py(name: str = "hip:mi300"):
    """Register ``name`` as a backend of an installed gt4py (>= 1.0, gtc based).  Returns the backend class.

    Prerequisite besides gt4py itself: **cupy for ROCm**.  gt4py converts every argument of a GPU backend with ``cp.asarray``
    before ``run()`` sees it (/root/reference/src/gt4py/cartesian/stencil_object.py:69-93 ->
    storage/cartesian/utils.py:176-215: ``device == "gpu"`` => cupy, asserted present) and allocates
    ``gt4py.storage.*(backend=name)`` with cupy (storage/cartesian/interface.py:84-100).  What arrives in ``run()`` is therefore
    a cupy array: this module reads it through ``__cuda_array_interface__`` (strides included) and needs nothing else of it."""
    try:
        from gt4py.cartesian import backend as gt4py_backend  # type: ignore
        from gt4py.cartesian.backend import base as gt4py_base  # type: ignore
        from gt4py.cartesian.gtc import passes as gtc_passes  # type: ignore
        from gt4py.cartesian.gtc.gtir_to_oir import GTIRToOIR  # type: ignore
    except Exception as ex:  # the compat shim of this repo also answers to `gt4py`: it has no `gtc`
        raise ImportError("register_with_gt4py() needs a real gt4py install (Python >= 3.12 with gt4py's "
                          "dependencies); in this image use gt4py_amd.cartesian.gtscript directly") from ex
    from ..storage import layout as amd_layout

    preset = amd_layout.from_name("hip:mi300")

    class HipMI300Backend(gt4py_base.BaseBackend):  # pragma: no cover - needs gt4py
        options = {"device_sync": {"versioning": True, "type": bool},
                   "use_kernel_library": {"versioning": True, "type": bool},
                   "oir_pipeline": {"versioning": True, "type": gtc_passes.OirPipeline}}
        # gt4py's LayoutInfo knows ONE alignment, in items of whatever dtype is allocated (storage/cartesian/interface.py:95-100);
        # the in-tree allocator aligns rows to 128 BYTES whatever the item size (storage/layout.py: `alignment_bytes`).  The
        # smallest item count that gives every 4- and 8-byte dtype rows on 128-byte boundaries is 32 -- fp32 rows then match the
        # in-tree layout exactly (128 B), fp64 rows get gt:gpu's 256 B (a multiple of 128: every kernel's fast path applies; the
        # measured cost against 128-byte rows is 1 % on the 512^3 Laplacian, profiles/r4_row_alignment.txt).  The preset's own
        # figure (16 items) would give fp32 rows 64 bytes: a different layout for the same backend name.
        storage_info = {"alignment": max(int(preset.get("alignment_bytes", 128)) // 4, int(preset["alignment"])), "device": "gpu",
                        "layout_map": preset["layout_map"], "is_optimal_layout": preset["is_optimal_layout"]}
        languages = {"computation": "hip", "bindings": ["python"]}

        def generate(self):
            self.check_options(self.builder.options)
            # the un-optimised OIR: one horizontal execution per statement (per `if`), which is the granularity of
            # this repo's own frontend; merging, inlining and caching are the planner's job here
            oir = GTIRToOIR().visit(self.builder.gtir)
            pipeline = self.builder.options.backend_opts.get("oir_pipeline")
            if pipeline is not None:
                oir = pipeline.run(oir)
            impl = stencil_from_oir(oir, name=self.builder.options.name,
                                    backend_opts={k: v for k, v in self.builder.options.backend_opts.items()
                                                  if k in ("device_sync", "use_kernel_library")})
            return _wrap_for_gt4py(self.builder, impl, gt4py_base)

        def load(self):
            return None  # kernels are cached by gt4py_amd (code objects on disk); the stencil class is rebuilt

    HipMI300Backend.name = name
    return gt4py_backend.register(HipMI300Backend)


def _wrap_for_gt4py(builder, impl, gt4py_base):  # pragma: no cover - needs gt4py
    """A gt4py ``StencilObject`` subclass whose ``run`` forwards to the gt4py_amd implementation: gt4py keeps doing
    argument normalisation and validation (stencil_object.py:531-612), this repo does the launches."""
    from gt4py.cartesian.backend.module_generator import make_args_data_from_gtir  # type: ignore
    from gt4py.cartesian.stencil_object import StencilObject  # type: ignore

    args_data = make_args_data_from_gtir(builder.gtir_pipeline)

    def run(self, _domain_, _origin_, exec_info=None, **kwargs):
        # what gt4py's `_call_run` hands over are cupy arrays (every argument went through `cp.asarray`, stencil_object.py:69-93,
        # 585-609): zero-copy views through `__cuda_array_interface__`, strides included -- the launch path wants DeviceArrays
        from ..storage.device_array import as_device_array

        fields = {k: (as_device_array(v) if v is not None and k in args_data.field_info else v) for k, v in kwargs.items()}
        impl.run(_domain_=tuple(_domain_), _origin_={k: tuple(v) for k, v in _origin_.items()}, exec_info=exec_info, **fields)

    def call(self, *args, domain=None, origin=None, validate_args=True, exec_info=None, **kwargs):
        bound = inspect.signature(builder.definition).bind_partial(*args, **kwargs)
        fields = {k: v for k, v in bound.arguments.items() if k in args_data.field_info}
        params = {k: v for k, v in bound.arguments.items() if k in args_data.parameter_info}
        self._call_run(field_args=fields, parameter_args=params, domain=domain, origin=origin,
                       validate_args=validate_args, exec_info=exec_info)

    attrs = {
        "_gt_id_": builder.stencil_id.version, "definition_func": staticmethod(builder.definition),
        "backend": property(lambda self: builder.backend.name), "source": property(lambda self: impl.source),
        "domain_info": property(lambda self: args_data.domain_info), "field_info": property(lambda self: args_data.field_info),
        "parameter_info": property(lambda self: args_data.parameter_info),
        "constants": property(lambda self: dict(builder.externals)), "options": property(lambda self: builder.options.as_dict()),
        "run": run, "__call__": call, "__module__": builder.module_qualname,
    }
    return type(builder.class_name, (StencilObject,), attrs)
