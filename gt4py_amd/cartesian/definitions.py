"""Run-time metadata types of a compiled stencil and the small tuple algebra they rely on.

Mirrors the *semantics* (names, attributes, comparison rules) of
/root/reference/src/gt4py/cartesian/definitions.py:59-112 (AccessKind, DomainInfo, FieldInfo,
ParameterInfo) and of the tuple classes in
/root/reference/src/gt4py/cartesian/gtc/definitions.py:141-460 (Index, Shape, Boundary) that
``StencilObject`` uses to normalise origins and validate domains.  Only what the call path needs is
implemented; this is not the compiler's IR toolkit.
"""

from __future__ import annotations

import enum
import functools
import numbers
import os
import platform
from dataclasses import dataclass
from typing import Any, Iterable, Sequence, Tuple

import numpy as np

AXES = ("I", "J", "K")

_ARCH_BITS = platform.architecture()[0][:2]
#: default width of untyped ``int`` literals (definitions.py:35-38 of the reference)
LITERAL_INT_PRECISION = int(os.environ.get("GT4PY_LITERAL_INT_PRECISION", default=_ARCH_BITS))
#: default width of untyped ``float`` literals (definitions.py:40-43 of the reference)
LITERAL_FLOAT_PRECISION = int(os.environ.get("GT4PY_LITERAL_FLOAT_PRECISION", default=_ARCH_BITS))


def get_integer_type(bits: int):
    try:
        return {8: np.int8, 16: np.int16, 32: np.int32, 64: np.int64}[bits]
    except KeyError:
        raise NotImplementedError("Unknown integer precision type") from None


def get_float_type(bits: int):
    try:
        return {32: np.float32, 64: np.float64}[bits]
    except KeyError:
        raise NotImplementedError("Unknown float precision type") from None


@enum.unique
class AccessKind(enum.IntFlag):
    NONE = 0
    READ = 1
    WRITE = 2
    READ_WRITE = READ | WRITE

    def __str__(self) -> str:
        return self.name


# --------------------------------------------------------------------------------------------
# masks
# --------------------------------------------------------------------------------------------
def filter_mask(seq: Sequence[Any], mask: Sequence[bool]) -> Tuple[Any, ...]:
    """Keep the entries of ``seq`` whose mask bit is set."""
    return tuple(v for v, m in zip(seq, mask) if m)


def interpolate_mask(seq: Iterable[Any], mask: Sequence[bool], default: Any) -> Tuple[Any, ...]:
    """Spread ``seq`` over the set bits of ``mask``, ``default`` elsewhere."""
    it = iter(seq)
    return tuple(next(it) if m else default for m in mask)


# --------------------------------------------------------------------------------------------
# numeric tuples
# --------------------------------------------------------------------------------------------
class _IntTuple(tuple):
    """Fixed-length tuple of integers with element-wise arithmetic and the reference's
    *partial-order* comparisons (gtc/definitions.py:141-179): ``a < b`` holds when no element is
    greater and at least one is smaller; ``a <= b`` when every element is <=."""

    __slots__ = ()
    _DEFAULT = 0
    _MIN = None

    def __new__(cls, values, *more):
        if more:
            values = (values, *more)
        values = tuple(values)
        for v in values:
            if not isinstance(v, numbers.Integral) or (cls._MIN is not None and v < cls._MIN):
                raise TypeError(f"Invalid {cls.__name__} definition")
        return super().__new__(cls, tuple(int(v) for v in values))  # numpy integer scalars -> int

    # constructors -------------------------------------------------------------------------
    @classmethod
    def zeros(cls, ndims: int = 3):
        return cls([0] * ndims)

    @classmethod
    def from_k(cls, value, ndims: int = 3):
        return cls([value] * ndims)

    @classmethod
    def from_value(cls, value):
        if isinstance(value, Iterable):
            return cls(list(value))
        return cls.from_k(value)

    @classmethod
    def from_mask(cls, seq, mask, default=None):
        return cls(interpolate_mask(seq, mask, cls._DEFAULT if default is None else default))

    # helpers ------------------------------------------------------------------------------
    def _coerce(self, other):
        if isinstance(other, numbers.Integral):
            return type(self)([other] * len(self))
        other = tuple(other)
        if len(other) != len(self):
            raise ValueError(f"Incompatible instance '{other}'")
        return other

    def _signs(self, other):
        return [(a > b) - (a < b) for a, b in zip(self, self._coerce(other))]

    def filter_mask(self, mask):
        return type(self)(filter_mask(self, mask))

    # arithmetic ---------------------------------------------------------------------------
    def __add__(self, other):
        return type(self)([a + b for a, b in zip(self, self._coerce(other))])

    def __sub__(self, other):
        return type(self)([a - b for a, b in zip(self, self._coerce(other))])

    def __and__(self, other):  # element-wise min ("intersection")
        return type(self)([min(a, b) for a, b in zip(self, self._coerce(other))])

    def __or__(self, other):  # element-wise max ("union")
        return type(self)([max(a, b) for a, b in zip(self, self._coerce(other))])

    # partial order ------------------------------------------------------------------------
    def __lt__(self, other):
        s = self._signs(other)
        return any(x < 0 for x in s) and not any(x > 0 for x in s)

    def __le__(self, other):
        return all(x <= 0 for x in self._signs(other))

    def __gt__(self, other):
        s = self._signs(other)
        return any(x > 0 for x in s) and not any(x < 0 for x in s)

    def __ge__(self, other):
        return all(x >= 0 for x in self._signs(other))

    def __eq__(self, other):
        try:
            return all(x == 0 for x in self._signs(other))
        except (ValueError, TypeError):
            return False

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return tuple.__hash__(self)

    def __repr__(self):
        return f"{type(self).__name__}({tuple.__repr__(self)})"

    def __str__(self):
        return tuple.__repr__(self)


class Index(_IntTuple):
    """Grid index (integers of any sign)."""

    __slots__ = ()


class Shape(_IntTuple):
    """Grid shape (integers >= 0)."""

    __slots__ = ()
    _DEFAULT = 1
    _MIN = 0


class Boundary(tuple):
    """Per-axis (lower, upper) halo widths around the compute domain."""

    __slots__ = ()

    def __new__(cls, ranges, *more):
        if more:
            ranges = (ranges, *more)
        ranges = tuple((int(lo), int(hi)) for lo, hi in ranges)
        return super().__new__(cls, ranges)

    @classmethod
    def zeros(cls, ndims: int = 3):
        return cls([(0, 0)] * ndims)

    @classmethod
    def from_offset(cls, offset):
        return cls([(-min(0, int(o)), max(0, int(o))) for o in offset])

    @property
    def lower_indices(self) -> Index:
        return Index([r[0] for r in self])

    @property
    def upper_indices(self) -> Index:
        return Index([r[1] for r in self])

    @property
    def frame_size(self) -> Shape:
        return Shape([r[0] + r[1] for r in self])

    def __or__(self, other):
        return Boundary([(max(a[0], b[0]), max(a[1], b[1])) for a, b in zip(self, other)])

    def __repr__(self):
        return f"Boundary({tuple.__repr__(self)})"


# --------------------------------------------------------------------------------------------
# stencil metadata
# --------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class DomainInfo:
    parallel_axes: Tuple[str, ...]
    sequential_axis: str
    min_sequential_axis_size: int
    ndim: int


@dataclass(frozen=True)
class FieldInfo:
    access: AccessKind
    boundary: Boundary
    axes: Tuple[str, ...]
    data_dims: Tuple[int, ...]
    dtype: np.dtype

    def __repr__(self):
        return (
            f"FieldInfo(access=AccessKind.{self.access.name}, boundary={self.boundary!r}, "
            f"axes={self.axes!r}, data_dims={self.data_dims!r}, dtype={self.dtype!r})"
        )

    @functools.cached_property
    def domain_mask(self) -> Tuple[bool, ...]:
        return tuple(axis in self.axes for axis in AXES)

    @functools.cached_property
    def domain_ndim(self) -> int:
        return len(self.axes)

    @functools.cached_property
    def mask(self) -> Tuple[bool, ...]:
        return (*self.domain_mask, *((True,) * len(self.data_dims)))

    @functools.cached_property
    def ndim(self) -> int:
        return len(self.axes) + len(self.data_dims)


@dataclass(frozen=True)
class ParameterInfo:
    access: AccessKind
    dtype: np.dtype

    def __repr__(self):
        return f"ParameterInfo(access=AccessKind.{self.access.name}, dtype={self.dtype!r})"


@dataclass
class BuildOptions:
    """Subset of the reference's BuildOptions (definitions.py:114-150) that influences the hot path."""

    name: str
    module: str
    backend_opts: dict
    build_info: Any = None
    rebuild: bool = False
    literal_int_precision: int = LITERAL_INT_PRECISION
    literal_float_precision: int = LITERAL_FLOAT_PRECISION
    impl_opts: Any = None

    @property
    def qualified_name(self) -> str:
        return ".".join([self.module, self.name])

    def as_dict(self) -> dict:
        return {
            "name": self.name,
            "module": self.module,
            "backend_opts": dict(self.backend_opts),
            "rebuild": self.rebuild,
            "literal_int_precision": self.literal_int_precision,
            "literal_float_precision": self.literal_float_precision,
            "_impl_opts": dict(self.impl_opts or {}),
        }


class GTError(Exception):
    pass


class GTScriptSyntaxError(GTError):
    def __init__(self, message, *, loc=None):
        super().__init__(message)
        self.loc = loc


class GTScriptSymbolError(GTScriptSyntaxError):
    pass


class GTScriptDefinitionError(GTScriptSyntaxError):
    pass


class GTScriptValueError(GTScriptDefinitionError):
    pass


class GTScriptDataTypeError(GTScriptSyntaxError):
    pass


class GTScriptAssertionError(GTError):
    """A ``compile_assert(...)`` whose condition is false (frontend/exceptions.py:89-98 of the reference)."""

    def __init__(self, source, *, loc=None):
        where = f" at line {loc}" if loc else ""
        super().__init__(f"Assertion failed{where}:\n{source}")
        self.loc = loc
