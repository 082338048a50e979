"""Test-only array carriers: objects that expose an array through the interface protocols and NOTHING else, optionally annotated
with ``__gt_dims__`` / ``__gt_origin__`` -- what the call interface must accept in place of a storage
(docs/user/cartesian/arrays.rst:25-37; the reference's tests use wrappers in the same role)."""

from __future__ import annotations

import dataclasses
from typing import Any, Optional, Sequence


@dataclasses.dataclass(frozen=True)
class _Carrier:
    """Keyword-only: ``array`` plus at most the two annotations.  The annotations are real attributes only when given (the
    call interface probes for them with ``getattr``)."""

    array: Any
    origin: Optional[Sequence[int]] = None
    dimensions: Optional[Sequence[str]] = None
    device_interface: bool = True  # expose __cuda_array_interface__ too (False: a host-only carrier)

    def __post_init__(self):
        rank = len(self.array.shape)
        for what, value in (("origin", self.origin), ("dimensions", self.dimensions)):
            if value is not None and len(value) != rank:
                raise ValueError(f"{what} {tuple(value)!r} does not have one entry per axis of an array of shape {tuple(self.array.shape)}")
        if self.origin is not None:
            object.__setattr__(self, "__gt_origin__", tuple(self.origin))
        if self.dimensions is not None:
            object.__setattr__(self, "__gt_dims__", self.dimensions)

    def __getattr__(self, name):  # (only reached for attributes that are not set)
        if name == "__array_interface__" or (name == "__cuda_array_interface__" and object.__getattribute__(self, "device_interface")):
            return getattr(object.__getattribute__(self, "array"), name)
        raise AttributeError(name)


def OriginWrapper(*, array, origin):
    return _Carrier(array=array, origin=origin)


def DimensionsWrapper(*, array, dimensions):
    return _Carrier(array=array, dimensions=dimensions)


def HostOnlyWrapper(array, origin=None, dimensions=None):
    """Exposes ONLY __array_interface__ (no __cuda_array_interface__), for host-side tests."""
    return _Carrier(array=array, origin=origin, dimensions=dimensions, device_interface=False)
