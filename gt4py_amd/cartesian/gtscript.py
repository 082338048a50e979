"""GTScript user surface: the ``@stencil`` decorator and the names stencil definitions use.

Mirrors the public names of /root/reference/src/gt4py/cartesian/gtscript.py (``stencil`` :219,
``lazy_stencil`` :403, ``Field`` :793, axes ``I/J/K/IJ/IJK`` :665-683, ``PARALLEL/FORWARD/BACKWARD``
:695-702, ``computation/interval`` :830-846, math builtins) so that existing stencil code is a drop-in.
As in the reference, the body of a definition is never executed: it is read with ``inspect`` and
parsed (frontend.py); the context managers and math functions below are stubs that only make the
names importable.
"""

from __future__ import annotations

import collections.abc
import inspect
import time
import types
from typing import Any, Callable, Dict, Optional

import numpy as np

from . import definitions as gt_definitions

# ---- axes ----------------------------------------------------------------------------------------


class ShiftedAxis:
    def __init__(self, name: str, shift: int):
        self.name, self.shift = name, shift

    def __repr__(self):
        return f"ShiftedAxis(name={self.name}, shift={self.shift})"


class AxisIndex:
    """``I[0] + 2``, ``K[-1]``: a position relative to the start (index >= 0) or the end (index < 0) of an axis,
    usable as an external for region and interval bounds (gtscript.py:557-590 of the reference)."""

    def __init__(self, axis: str, index: int, offset: int = 0):
        self.axis, self.index, self.offset = axis, index, offset

    def __repr__(self):
        return f"AxisIndex(axis={self.axis}, index={self.index}, offset={self.offset})"

    def __str__(self):
        return f"{self.axis}[{self.index}] + {self.offset}"

    def __eq__(self, other):
        return repr(self) == repr(other)

    def __hash__(self):
        return hash(repr(self))

    def __add__(self, offset):
        import numbers

        if not isinstance(offset, numbers.Integral):
            raise TypeError("Offset should be an integer type")
        return self if offset == 0 else AxisIndex(self.axis, self.index, self.offset + int(offset))

    __radd__ = __add__

    def __sub__(self, offset):
        return self.__add__(-offset)

    def __rsub__(self, offset):
        return self.__add__(-offset)


class AxisInterval:
    def __init__(self, axis: str, start: int, end: int):
        assert start < end
        self.axis, self.start, self.end = axis, start, end

    def __repr__(self):
        return f"AxisInterval(axis={self.axis}, start={self.start}, end={self.end})"

    def __len__(self):
        return self.end - self.start


class Axis:
    """A cartesian axis symbol; ``Axis + n`` spells a field offset (``f[I + 1]``), ``Axis[n]`` an ``AxisIndex``."""

    def __getitem__(self, interval):
        if isinstance(interval, slice):
            return AxisInterval(self.name, interval.start, interval.stop)
        if isinstance(interval, int):
            return AxisIndex(self.name, interval)
        raise TypeError("Unrecognized index type")

    def __init__(self, name: str):
        assert name
        self.name = name
        self.__gt_axis_name__ = name

    def __repr__(self):
        return f"Axis(name={self.name})"

    def __str__(self):
        return self.name

    def __add__(self, shift):
        if not isinstance(shift, int):
            raise TypeError(f"Can only add type int, got {type(shift)}")
        return ShiftedAxis(self.name, shift)

    def __sub__(self, shift):
        if not isinstance(shift, int):
            raise TypeError(f"Can only subtract type int, got {type(shift)}")
        return ShiftedAxis(self.name, -shift)


I = Axis("I")  # noqa: E741
J = Axis("J")
K = Axis("K")
IJ = (I, J)
IK = (I, K)
JK = (J, K)
IJK = (I, J, K)

# ---- iteration orders ----------------------------------------------------------------------------
FORWARD = +1
BACKWARD = -1
PARALLEL = 0

_VALID_DATA_TYPES = (bool, np.bool_, int, np.int8, np.int16, np.int32, np.int64, float, np.float32, np.float64)


class _FieldDescriptor:
    """What ``Field[...]`` evaluates to in an annotation: dtype (or a string key resolved through
    ``dtypes=``), axes and data dimensions."""

    def __init__(self, dtype, axes, data_dims=()):
        if isinstance(dtype, str):
            self.dtype = dtype
        else:
            try:
                dt = np.dtype(dtype)
            except TypeError as ex:
                raise ValueError("Invalid data type descriptor") from ex
            if dt.shape:
                assert not data_dims
                data_dims = dt.shape
                dt = dt.base
            if dt not in [np.dtype(t) for t in _VALID_DATA_TYPES]:
                raise ValueError("Invalid data type descriptor")
            self.dtype = dt
        self.axes = tuple(axes) if isinstance(axes, collections.abc.Collection) else (axes,)
        if data_dims and not isinstance(data_dims, collections.abc.Collection):
            data_dims = (data_dims,)
        self.data_dims = tuple(data_dims)

    def __repr__(self):
        return f"_FieldDescriptor(dtype={self.dtype!r}, axes={self.axes!r}, data_dims={self.data_dims!r})"


class _FieldDescriptorMaker:
    @staticmethod
    def _is_axes_spec(spec) -> bool:
        return isinstance(spec, Axis) or (
            isinstance(spec, collections.abc.Collection) and not isinstance(spec, str)
            and len(spec) > 0 and all(isinstance(i, Axis) for i in spec)
        )

    def __getitem__(self, field_spec):
        axes = IJK
        data_dims = ()
        if isinstance(field_spec, str) or not isinstance(field_spec, collections.abc.Collection):
            dtype = field_spec  # Field[dtype]
        elif self._is_axes_spec(field_spec[0]):
            assert len(field_spec) == 2  # Field[axes, dtype]
            axes, dtype = field_spec
        elif len(field_spec) == 2 and not self._is_axes_spec(field_spec[1]):
            dtype = field_spec  # Field[(dtype, data_dims)]
        else:
            raise ValueError("Invalid field type descriptor")
        if isinstance(dtype, collections.abc.Collection) and not isinstance(dtype, str):
            assert len(dtype) == 2
            dtype, data_dims = dtype
        return _FieldDescriptor(dtype, axes, data_dims)


Field = _FieldDescriptorMaker()


class _GlobalTableDescriptorMaker(_FieldDescriptorMaker):
    """``GlobalTable[(dtype, (n0, n1, ...))]``: a read-only array with data dimensions only, indexed
    absolutely through ``table.A[i0, i1, ...]`` (gtscript.py:782-796 of the reference)."""

    def __getitem__(self, field_spec):
        if not isinstance(field_spec, collections.abc.Collection) or len(field_spec) != 2:
            raise ValueError("GlobalTable is defined by a tuple (type, [axes_size..])")
        dtype, data_dims = field_spec
        return _FieldDescriptor(dtype, [], data_dims)


GlobalTable = _GlobalTableDescriptorMaker()


# ---- statements that only exist syntactically -------------------------------------------------
class _NullContext:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


def computation(order):
    return _NullContext()


def interval(*args):
    return _NullContext()


def horizontal(*args):
    return _NullContext()


class _Region:
    def __getitem__(self, *args):
        pass


region = _Region()


def externals(*args):
    return args


def __INLINED(compile_if_expression):
    pass


def compile_assert(expr):
    pass


__externals__ = "Placeholder"
__gtscript__ = "Placeholder"

int32, int64, float32, float64 = np.int32, np.int64, np.float32, np.float64

MATH_BUILTINS = (
    "abs", "min", "max", "mod", "sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh",
    "asinh", "acosh", "atanh", "sqrt", "exp", "log", "log10", "gamma", "cbrt", "isfinite", "isinf",
    "isnan", "floor", "ceil", "trunc", "erf", "erfc", "round", "round_away_from_zero",
)


def _math_stub(name):
    def stub(*args):
        raise RuntimeError(f"gtscript.{name} is only meaningful inside a stencil definition")

    stub.__name__ = name
    return stub


for _name in MATH_BUILTINS:  # abs / min / max / round shadow the builtins in this module, as in the reference
    globals()[_name] = _math_stub(_name)


ENUM_REGISTER: Dict[str, type] = {}


def enum(class_):
    """Mark an ``IntEnum`` subclass as readable inside stencils: ``MyEnum.A`` becomes its integer value and
    a parameter annotated with the enum is an integer scalar (gtscript.py:163-168, gtscript_frontend.py:2457-2468)."""
    import enum as _enum

    name = class_.__name__
    if name in ENUM_REGISTER:
        raise ValueError(f"Enum names must be unique. @gtscript.enum {name} is already taken.")
    if not (isinstance(class_, type) and issubclass(class_, _enum.IntEnum)):
        raise ValueError(f"Enum {name} needs to derive from `enum.IntEnum`.")
    ENUM_REGISTER[name] = class_
    return class_


def function(func):
    """``@gtscript.function`` marker (inlined by the frontend when called from a stencil).  Like the reference's
    ``annotate_definition`` it records what the names of the enclosing scope mean NOW
    (gtscript_frontend.py:2121-2267): a constant rebound later does not change the function."""
    from . import call_inliner

    setattr(func, "__gtscript_context__", call_inliner._context_of(func))
    setattr(func, "__gtscript_function__", True)
    return func


def lazy_function(*, before_annotation=None, after_annotation=None):
    """Mark a GTScript function that is only annotated right before its first use; the hooks run before / after
    (gtscript.py:179-215 of the reference)."""

    def wrapper(func):
        def inner_function():
            if before_annotation is not None:
                before_annotation(func)
            function(func)
            if after_annotation is not None:
                after_annotation(func)
            return func

        return inner_function

    return wrapper


# ---- decorators --------------------------------------------------------------------------------
def _resolve_annotations(definition: Callable, dtypes: Dict[Any, Any]) -> Dict[str, Any]:
    """Evaluated annotations of ``definition`` with ``dtypes`` substitutions applied
    (counterpart of ``_set_arg_dtypes``, gtscript.py:132-168 of the reference)."""
    sig = inspect.signature(definition)
    out: Dict[str, Any] = {}
    for pname, param in sig.parameters.items():
        ann = param.annotation
        if ann is inspect.Parameter.empty:
            raise gt_definitions.GTScriptDefinitionError(
                f"Missing type annotation for argument '{pname}' of '{definition.__name__}'"
            )
        if isinstance(ann, str):
            if ann in dtypes:
                ann = dtypes[ann]
            else:
                scope = dict(vars(inspect.getmodule(definition) or types.ModuleType("_")))
                scope.update(getattr(definition, "__globals__", {}))
                scope.setdefault("Field", Field)
                scope.update({"I": I, "J": J, "K": K, "IJ": IJ, "IK": IK, "JK": JK, "IJK": IJK, "np": np})
                try:
                    ann = eval(ann, scope, dict(dtypes))  # noqa: S307 - annotation strings of the user's own code
                except Exception as ex:
                    raise gt_definitions.GTScriptDefinitionError(
                        f"Cannot evaluate annotation '{param.annotation}' of argument '{pname}'"
                    ) from ex
        if isinstance(ann, _FieldDescriptor):
            if isinstance(ann.dtype, str) or (not isinstance(ann.dtype, np.dtype)):
                if ann.dtype in dtypes:
                    ann = _FieldDescriptor(dtypes[ann.dtype], ann.axes, ann.data_dims)
                else:
                    raise gt_definitions.GTScriptDefinitionError(
                        f"Unknown dtype key '{ann.dtype}' for argument '{pname}' (pass dtypes={{...}})"
                    )
        elif ann in dtypes:
            ann = dtypes[ann]
        out[pname] = ann
    return out


def stencil(backend, definition=None, *, build_info=None, dtypes=None, externals=None, format_source=True,
            name=None, rebuild=False, cache_settings=None, raise_if_not_cached=False,
            literal_int_precision=gt_definitions.LITERAL_INT_PRECISION,
            literal_float_precision=gt_definitions.LITERAL_FLOAT_PRECISION, **kwargs):
    """Build a stencil object from a definition for ``backend``; decorator or plain call.

    Same signature as the reference's ``gtscript.stencil`` (gtscript.py:219-400).  Unknown keyword
    arguments are backend options (validated by the backend, unknown ones warn); keys starting with
    ``_`` go to ``_impl_opts``.
    """
    from . import loader

    if build_info is not None and not isinstance(build_info, dict):
        raise ValueError(f"Invalid 'build_info' dictionary ('{build_info}')")
    if dtypes is not None and not isinstance(dtypes, dict):
        raise ValueError(f"Invalid 'dtypes' dictionary ('{dtypes}')")
    if externals is not None and not isinstance(externals, dict):
        raise ValueError(f"Invalid 'externals' dictionary ('{externals}')")
    if name is not None and not isinstance(name, str):
        raise ValueError(f"Invalid 'name' string ('{name}')")
    if not isinstance(rebuild, bool):
        raise ValueError(f"Invalid 'rebuild' bool value ('{rebuild}')")

    module = None
    if name:
        parts = name.split(".")
        name = parts[-1]
        module = ".".join(parts[:-1])
    caller = inspect.currentframe().f_back
    module = module or (caller.f_globals.get("__name__", "__main__") if caller else "__main__")

    impl_opts = {k: v for k, v in kwargs.items() if k.startswith("_")}
    backend_opts = {k: v for k, v in kwargs.items() if not k.startswith("_")}
    if build_info is not None:
        build_info.update({k: 0.0 for k in ("parse_time", "module_time", "codegen_time", "build_time", "load_time")})

    def _decorator(definition_func):
        if not isinstance(definition_func, types.FunctionType):
            if hasattr(definition_func, "definition_func"):  # a StencilObject
                definition_func = definition_func.definition_func
            elif callable(definition_func):
                definition_func = definition_func.__call__
        options = gt_definitions.BuildOptions(
            name=name or definition_func.__name__,
            module=module,
            backend_opts=backend_opts,
            build_info=build_info,
            rebuild=rebuild,
            literal_int_precision=literal_int_precision,
            literal_float_precision=literal_float_precision,
            impl_opts=impl_opts,
        )
        return loader.gtscript_loader(definition_func, backend=backend, build_options=options,
                                      externals=externals or {}, dtypes=dtypes or {})

    return _decorator if definition is None else _decorator(definition)


def lazy_stencil(backend, definition=None, *, eager=False, **stencil_kwargs):
    """Deferred build: returns an object whose first call builds the stencil
    (gtscript.py:403-520 of the reference; ``eager=True`` builds immediately)."""

    class _Lazy:
        def __init__(self, func):
            self._func = func
            self._impl = None

        @property
        def implementation(self):
            if self._impl is None:
                self._impl = stencil(backend, self._func, **stencil_kwargs)
            return self._impl

        def __call__(self, *args, **kw):
            return self.implementation(*args, **kw)

        def __getattr__(self, item):
            return getattr(self.implementation, item)

    def _decorator(func):
        lazy = _Lazy(func)
        if eager:
            lazy.implementation  # noqa: B018
        return lazy

    return _decorator if definition is None else _decorator(definition)


__all__ = [
    "Axis", "AxisIndex", "BACKWARD", "FORWARD", "Field", "GlobalTable", "enum", "I", "IJ", "IJK", "IK", "J", "JK", "K", "PARALLEL",
    "__INLINED", "__externals__", "__gtscript__", "compile_assert", "computation", "externals",
    "float32", "float64", "function", "lazy_function", "horizontal", "int32", "int64", "interval", "lazy_stencil",
    "region", "stencil",
]
