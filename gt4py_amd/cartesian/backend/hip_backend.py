"""``hip:mi300`` -- the MI355X-native stencil backend.

Registered through the reference's own plug-in mechanism (``@register`` on a ``Backend`` subclass
with ``name / options / storage_info / languages`` --
/root/reference/src/gt4py/cartesian/backend/base.py:35-152; GPU precedent ``GTGpuBackend``,
backend/gtcpp_backend.py:169-183).  Where ``gt:gpu`` generates GridTools C++ and JIT-compiles a
pybind11 module, this backend first tries to recognise the stencil (hip_templates.py) and bind it to a
hand-written gfx950 kernel in ``libgt4py_amd.so`` through the C ABI (include/gt4py_amd.h); any other
stencil of the accepted sub-language goes to the generic executor (hip_codegen.py -> HIP source ->
hiprtc inside the library -> one launch per stage, hip_generic.py).

There is no CPU fallback: what neither path can run exactly raises ``NotImplementedError`` at
decoration time; a missing shared library raises ``RuntimeError``.
"""

from __future__ import annotations

import ctypes
import inspect
import weakref
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple, Type

import numpy as np

from . import base, hip_codegen, hip_generic, hip_templates
from .. import definitions as gt_definitions, frontend, ir
from ..stencil_object import StencilObject
from ... import _lib
from ...storage.layout import HIP_MI300_LAYOUT


# ---------------------------------------------------------------------------------------------
# alpha-equivalence of typed stencil IR
# ---------------------------------------------------------------------------------------------
def canonical_form(stencil: ir.Stencil) -> Tuple[Tuple, Dict[str, str]]:
    """Structure of ``stencil`` with symbols renamed by order of first appearance.

    Returns (hashable structure, {canonical name -> original name}).  Canonical names carry the
    symbol kind: ``f<n>`` API field, ``t<n>`` temporary, ``p<n>`` scalar parameter.
    """
    kinds = {d.name: "f" for d in stencil.fields}
    kinds.update({d.name: "t" for d in stencil.temporaries})
    kinds.update({d.name: "p" for d in stencil.params})
    rename: Dict[str, str] = {}

    def canon(name: str) -> str:
        if name not in rename:
            kind = kinds[name]
            rename[name] = f"{kind}{sum(1 for v in rename.values() if v[0] == kind)}"
        return rename[name]

    def expr(e: ir.Expr):
        if isinstance(e, ir.Literal):
            return ("lit", repr(e.value), str(e.dtype))
        if isinstance(e, ir.FieldAccess):
            return ("field", canon(e.name), e.offset, str(e.dtype), expr(e.koffset) if e.koffset is not None else None,
                    tuple(expr(d) if isinstance(d, ir.Expr) else d for d in e.data_index or ()), e.absolute_k)
        if isinstance(e, ir.AxisIndex):
            return ("axis", e.axis, str(e.dtype))
        if isinstance(e, ir.ScalarAccess):
            return ("scalar", canon(e.name), str(e.dtype))
        if isinstance(e, ir.UnaryOp):
            return ("un", e.op, expr(e.expr), str(e.dtype))
        if isinstance(e, ir.BinaryOp):
            return ("bin", e.op, expr(e.left), expr(e.right), str(e.dtype))
        if isinstance(e, ir.TernaryOp):
            return ("tern", expr(e.cond), expr(e.true_expr), expr(e.false_expr), str(e.dtype))
        if isinstance(e, ir.Cast):
            return ("cast", expr(e.expr), str(e.dtype))
        if isinstance(e, ir.NativeCall):
            return ("call", e.func, tuple(expr(a) for a in e.args), str(e.dtype))
        raise TypeError(e)

    comps = []
    for comp in stencil.computations:
        blocks = []
        for block in comp.blocks:
            body = []
            for stmt in block.body:
                mask = expr(stmt.mask) if stmt.mask is not None else None
                value = expr(stmt.value)  # right-hand side first, then the target
                loops = tuple((lid, expr(c)) for lid, c in stmt.loops)
                body.append((expr(stmt.target), value, mask, stmt.group, stmt.region, loops))
            iv = block.interval
            blocks.append(((iv.start.level.value, iv.start.offset, iv.end.level.value, iv.end.offset), tuple(body)))
        comps.append((comp.order.value, tuple(blocks)))
    return tuple(comps), {v: k for k, v in rename.items()}


@dataclass(frozen=True)
class KernelBinding:
    """A recognised stencil: which kernel family/variant, and which user symbol plays which role."""

    family: str  # "lap5" | "hdiff" | "tridiag"
    template: str  # template function name
    roles: Dict[str, str]  # template symbol -> user symbol
    dtype: np.dtype  # field dtype (float32 / float64)
    flags: int
    variant: int = 0


_LAP_VARIANTS = {"lap_notebook": _lib.LAP_NOTEBOOK, "lap_docs": _lib.LAP_DOCS, "lap_suite": _lib.LAP_SUITE,
                 "lap_avg": _lib.LAP_AVG}
_HDIFF_TEMPLATES = {"hdiff_limiter_field": (True, True), "hdiff_limiter_scalar": (True, False),
                    "hdiff_plain_field": (False, True), "hdiff_plain_scalar": (False, False)}
_TEMPLATE_CACHE: Dict[Tuple, Tuple[Tuple, Dict[str, str]]] = {}


def _template_form(name: str, T: np.dtype, S: Optional[np.dtype], options: gt_definitions.BuildOptions):
    key = (name, str(T), str(S), options.literal_int_precision, options.literal_float_precision)
    if key not in _TEMPLATE_CACHE:
        tmpl_options = gt_definitions.BuildOptions(
            name=name, module=hip_templates.__name__, backend_opts={},
            literal_int_precision=options.literal_int_precision,
            literal_float_precision=options.literal_float_precision,
        )
        dtypes = {"T": T}
        if S is not None:
            dtypes["S"] = S
        tmpl_ir = frontend.parse_stencil(getattr(hip_templates, name), externals={}, dtypes=dtypes, options=tmpl_options)
        _TEMPLATE_CACHE[key] = canonical_form(tmpl_ir)
    return _TEMPLATE_CACHE[key]


def recognise(stencil: ir.Stencil, options: gt_definitions.BuildOptions) -> Optional[KernelBinding]:
    """Bind ``stencil`` to a kernel family, or None when it is not one of the supported shapes."""
    user_form, user_names = canonical_form(stencil)
    used_fields = [d for d in stencil.fields if d.name in user_names.values()]
    if not used_fields:
        return None
    T = np.dtype(used_fields[0].dtype)
    if T not in (np.dtype("float32"), np.dtype("float64")):
        return None
    if any(np.dtype(d.dtype) != T or tuple(d.axes) != ("I", "J", "K") or d.data_dims for d in used_fields):
        return None
    used_params = [p for p in stencil.params if p.name in user_names.values()]
    S = np.dtype(used_params[0].dtype) if used_params else None
    lit32 = options.literal_float_precision == 32

    candidates: List[str] = list(_LAP_VARIANTS) + ["tridiagonal_solver"]
    candidates += [n for n, (_, is_field) in _HDIFF_TEMPLATES.items() if is_field or S is not None]
    for name in candidates:
        needs_scalar = name in ("hdiff_limiter_scalar", "hdiff_plain_scalar")
        if needs_scalar != (S is not None):
            continue
        if S is not None and S not in (np.dtype("float32"), np.dtype("float64")):
            continue
        tmpl_form, tmpl_names = _template_form(name, T, S if needs_scalar else None, options)
        if tmpl_form != user_form:
            continue
        roles = {tmpl_names[c]: user_names[c] for c in tmpl_names}
        if name in _LAP_VARIANTS:
            flags = _lib.LAP_LITERAL_F32 if (lit32 and T == np.dtype("float32")) else 0
            return KernelBinding("lap5", name, roles, T, flags, _LAP_VARIANTS[name])
        if name in _HDIFF_TEMPLATES:
            limiter, _ = _HDIFF_TEMPLATES[name]
            flags = _lib.HDIFF_LIMITER if limiter else 0
            if lit32 and T == np.dtype("float32"):
                flags |= _lib.HDIFF_INTERNAL_F32
            if S == np.dtype("float32"):
                flags |= _lib.HDIFF_COEFF_F32
            return KernelBinding("hdiff", name, roles, T, flags)
        return KernelBinding("tridiag", name, roles, T, 0)
    return None


# ---------------------------------------------------------------------------------------------
# the generated StencilObject subclass
# ---------------------------------------------------------------------------------------------
class HipStencilObject(StencilObject):
    """StencilObject whose ``run`` launches a gfx950 kernel through the C ABI."""

    _gt_binding_: KernelBinding
    _gt_device_sync_: bool
    _gt_launch_cache_: Dict[Any, Any]

    def _run_implementation(self, domain, origin, exec_info, arguments: Dict[str, Any]) -> None:
        cls = type(self)
        binding = cls._gt_binding_
        lib = _lib.load()
        try:
            import torch

            stream = torch.cuda.current_stream().cuda_stream
        except Exception as ex:  # pragma: no cover - no GPU
            raise RuntimeError("hip:mi300 needs PyTorch-ROCm with a visible MI355X") from ex

        # ctypes structs of a call are cached per (arrays, origins, domain): building them costs more Python
        # time than the launch.  Arrays are remembered by identity through weak references.
        names = [binding.roles[r] for r in binding.roles if binding.roles[r] in cls._gt_field_info_]
        ckey = (tuple(int(d) for d in domain), tuple(id(arguments[n]) for n in names), tuple(tuple(origin[n]) for n in names))
        entry = cls._gt_launch_cache_.get(ckey)
        if entry is not None and all(r() is arguments[n] for n, r in entry[0]):
            _, dom, refs = entry
        else:
            refs = {}
            for role, name in binding.roles.items():
                if name in cls._gt_field_info_:
                    arr = arguments[name]
                    refs[role] = ctypes.byref(_lib.Field.make(arr.ptr, arr.shape, arr.strides, origin[name]))
            dom = _lib.domain3(domain)
            try:
                weak = [(n, weakref.ref(arguments[n])) for n in names]
                if len(cls._gt_launch_cache_) >= 8:
                    cls._gt_launch_cache_.pop(next(iter(cls._gt_launch_cache_)))
                cls._gt_launch_cache_[ckey] = (weak, dom, refs)
            except TypeError:  # no weak-reference support: do not cache
                pass
        fld = refs.__getitem__
        info = _lib.ExecInfo() if exec_info is not None else None
        info_ref = ctypes.byref(info) if info is not None else None
        suffix = "f64" if binding.dtype == np.dtype("float64") else "f32"
        if binding.family == "lap5":
            fname = f"gt4mi_lap5_{suffix}"
            rc = getattr(lib, fname)(dom, fld("inp"), fld("out"), binding.variant, binding.flags, stream, info_ref)
        elif binding.family == "hdiff":
            fname = f"gt4mi_hdiff_{suffix}"
            if "coeff" in binding.roles and binding.roles["coeff"] in cls._gt_field_info_:
                rc = getattr(lib, fname)(dom, fld("in_field"), fld("out_field"), fld("coeff"), 0.0,
                                         binding.flags, stream, info_ref)
            else:
                rc = getattr(lib, fname)(dom, fld("in_field"), fld("out_field"), None,
                                         float(arguments[binding.roles["coeff"]]), binding.flags, stream, info_ref)
        elif binding.family == "tridiag":
            fname = f"gt4mi_tridiag_{suffix}"
            rc = getattr(lib, fname)(dom, fld("inf"), fld("diag"), fld("sup"), fld("rhs"), fld("out"), stream, info_ref)
        else:  # pragma: no cover
            raise RuntimeError(f"unknown kernel family {binding.family}")
        _lib.check(fname, rc)
        if cls._gt_device_sync_:
            _lib.check("gt4mi_stream_sync", lib.gt4mi_stream_sync(stream))
        if exec_info is not None:
            exec_info["run_cpp_start_time"] = info.run_cpp_start_time
            exec_info["run_cpp_end_time"] = info.run_cpp_end_time
            if info.run_hip_end_time > 0.0:  # device-side interval of this call's kernels (hipEvent pair)
                exec_info["run_hip_start_time"] = info.run_hip_start_time
                exec_info["run_hip_end_time"] = info.run_hip_end_time


@base.register
class HipMI300Backend(base.BaseBackend):
    """MI355X (gfx950) backend: hand-written HIP kernels behind the gt4py.cartesian API."""

    name = "hip:mi300"
    options = {
        # same meaning as gt:gpu's option (backend/gtcpp_backend.py:178): synchronise after each call
        "device_sync": {"versioning": True, "type": bool},
        # False: skip the hand-written kernel library and always generate code (testing, comparisons)
        "use_kernel_library": {"versioning": True, "type": bool},
        # `while` loops: "statementwise" (default: the reference's numpy backend, the oracle of this path) or "pointwise" (its
        # compiled backends, gt:gpu among them); see frontend._Parser._pointwise_while
        "while_loops": {"versioning": True, "type": str},
    }
    storage_info = HIP_MI300_LAYOUT
    languages = {"computation": "hip", "bindings": ["c-abi/ctypes"]}

    def make_stencil_class(self) -> Type[StencilObject]:
        builder = self.builder
        binding = None
        if builder.options.backend_opts.get("use_kernel_library", True):
            binding = recognise(builder.stencil_ir, builder.options)
        program = None
        if binding is None:
            try:
                program = hip_codegen.generate(builder.stencil_ir)
            except hip_codegen.UnsupportedStencil as ex:
                supported = ", ".join(list(_LAP_VARIANTS) + list(_HDIFF_TEMPLATES) + ["tridiagonal_solver"])
                raise NotImplementedError(
                    f"Stencil '{builder.options.name}' is neither one of the hand-written gfx950 kernel families "
                    f"of backend 'hip:mi300' ({supported}; gt4py_amd/cartesian/backend/hip_templates.py) nor "
                    f"within reach of its generic executor: {ex}. There is no CPU fallback in this backend."
                ) from ex
        sig = inspect.signature(builder.definition)
        sig = sig.replace(parameters=[p.replace(annotation=inspect.Parameter.empty) for p in sig.parameters.values()])
        attrs = {
            "_gt_id_": builder.stencil_id,
            "definition_func": staticmethod(builder.definition),
            "_gt_backend_": self.name,
            "_gt_source_": builder.source,
            "_gt_domain_info_": builder.args_data.domain_info,
            "_gt_field_info_": builder.args_data.field_info,
            "_gt_parameter_info_": builder.args_data.parameter_info,
            "_gt_constants_": dict(builder.externals),
            "_gt_options_": builder.options.as_dict(),
            "_gt_signature_": sig,
            "_gt_device_sync_": bool(builder.options.backend_opts.get("device_sync", True)),
            "__module__": builder.options.module,
            "__doc__": builder.definition.__doc__,
        }
        if binding is not None:
            attrs["_gt_binding_"] = binding
            attrs["_gt_launch_cache_"] = {}
            return type(builder.class_name, (HipStencilObject,), attrs)
        attrs.update({"_gt_program_": program, "_gt_variants_": {}, "_gt_scratch_": {}, "_gt_launch_cache_": {}})
        return type(builder.class_name, (hip_generic.HipGenericStencilObject,), attrs)
