"""``gt4py_amd.storage`` -- allocators with per-backend layout, padding and alignment.

Same public surface as ``gt4py.storage`` (/root/reference/src/gt4py/storage/__init__.py and
storage/cartesian/interface.py:40-327): ``empty``, ``zeros``, ``ones``, ``full``, ``from_array``,
all keyword-only in ``backend=``, ``aligned_index=``, ``dimensions=``; plus ``register`` /
``from_name`` of the layout registry.  CPU presets return ``numpy.ndarray``; the ``hip:mi300``
preset returns a :class:`DeviceArray` in HBM.
"""

from __future__ import annotations

import collections.abc
import numbers
from typing import Optional, Sequence, Tuple

import numpy as np

from . import allocators, layout as layout_registry
from .device_array import DeviceArray, as_device_array, asnumpy
from .layout import LayoutInfo, from_name, layout_checker_factory, layout_maker_factory, register

__all__ = [
    "DeviceArray",
    "LayoutInfo",
    "asnumpy",
    "as_device_array",
    "empty",
    "from_array",
    "from_name",
    "full",
    "layout_checker_factory",
    "layout_maker_factory",
    "ones",
    "register",
    "zeros",
]


def _error_on_invalid_preset(backend) -> None:
    if backend not in layout_registry.REGISTRY:
        raise RuntimeError(f"Storage preset '{backend}' is not registered.")


def normalize_storage_spec(aligned_index, shape, dtype, dimensions):
    """Validate and normalise (aligned_index, shape, dtype, dimensions).

    Behaviour and error classes follow storage/cartesian/utils.py:84-165 of the reference
    (pinned by tests/storage_tests/unit_tests/test_interface.py:241-346): default dimensions are
    ``"IJK"[:ndim]`` (plus data dims), sub-array dtypes append data dimensions.
    """
    if shape is None or not (isinstance(shape, collections.abc.Sequence)
                             and all(isinstance(s, numbers.Integral) for s in shape)):
        raise TypeError("shape must be an iterable of ints.")
    if dimensions is None:
        n = len(shape)
        dimensions = list("IJK"[:n]) if n <= 3 else list("IJK") + [str(d) for d in range(n - 3)]
    if aligned_index is None:
        aligned_index = [0] * len(shape)

    dimensions = tuple(getattr(d, "__gt_axis_name__", d) for d in dimensions)
    if not all(isinstance(d, str) and (d.isdigit() or d in "IJK") for d in dimensions):
        raise ValueError(f"Invalid dimensions definition: '{dimensions}'")
    dimensions = tuple(str(d) for d in dimensions)

    if len(shape) != len(dimensions):
        raise ValueError(
            f"Dimensions ({dimensions}) and shape ({shape}) have non-matching sizes."
            f"len(shape)(={len(shape)}) must be equal to len(dimensions)(={len(dimensions)})."
        )
    shape = tuple(int(s) for s in shape)
    if any(s <= 0 for s in shape):
        raise ValueError(f"shape ({shape}) contains non-positive value.")

    if not (isinstance(aligned_index, collections.abc.Sequence)
            and all(isinstance(i, numbers.Integral) for i in aligned_index)):
        raise TypeError("aligned_index must be an iterable of ints.")
    if len(aligned_index) != len(shape):
        raise ValueError(
            f"Shape ({shape}) and aligned_index ({aligned_index}) have non-matching sizes."
            f"len(aligned_index)(={len(aligned_index)}) must be equal to len(shape)(={len(shape)})."
        )
    aligned_index = tuple(int(i) for i in aligned_index)
    if any(i < 0 for i in aligned_index):
        raise ValueError(f"aligned_index ({aligned_index}) contains negative value.")

    dtype = np.dtype(dtype)
    if dtype.shape:
        sub_dtype, sub_shape = dtype.subdtype
        aligned_index = (*aligned_index, *((0,) * dtype.ndim))
        shape = (*shape, *sub_shape)
        dimensions = (*dimensions, *(str(d) for d in range(dtype.ndim)))
        dtype = sub_dtype
    return aligned_index, shape, dtype, dimensions


def empty(shape: Sequence[int], dtype=np.float64, *, backend: str,
          aligned_index: Optional[Sequence[int]] = None, dimensions: Optional[Sequence[str]] = None,
          memory_class: Optional[int] = None):
    """Uninitialised array with the optimal strides/alignment for ``backend``
    (interface.py:40-102 of the reference).

    ``memory_class`` (0 / 1 / None; an extension, device presets only): the memory class a big field should live in, as
    ``stencil.placement_hint()`` suggests it from what the stencil does with the field (``placement.py``); None leaves the
    choice to the allocator's balance of live bytes."""
    _error_on_invalid_preset(backend)
    info = layout_registry.from_name(backend)
    allocate = allocators.allocate_gpu if info["device"] == "gpu" else allocators.allocate_cpu
    aligned_index, shape, dtype, dimensions = normalize_storage_spec(aligned_index, shape, dtype, dimensions)
    layout_map = info["layout_map"](dimensions)
    assert allocators.is_valid_layout_map(layout_map)
    # (a preset may give its alignment in BYTES -- hip:mi300: one HBM sector whatever the item size --; else the reference's rule:
    # `alignment` items of the array's dtype, interface.py:95-100)
    alignment_bytes = info.get("alignment_bytes") or info["alignment"] * dtype.itemsize
    alignment_bytes = -(-int(alignment_bytes) // dtype.itemsize) * dtype.itemsize
    if memory_class is not None and info["device"] == "gpu":
        if memory_class not in (0, 1):
            raise ValueError(f"memory_class must be 0, 1 or None, not {memory_class!r}")
        from . import placement

        with placement.want(int(memory_class)):
            _, array = allocate(shape, layout_map, dtype, alignment_bytes, aligned_index)
    else:
        _, array = allocate(shape, layout_map, dtype, alignment_bytes, aligned_index)
    return array


def full(shape: Sequence[int], fill_value, dtype=np.float64, *, backend: str,
         aligned_index: Optional[Sequence[int]] = None, dimensions: Optional[Sequence[str]] = None,
         memory_class: Optional[int] = None):
    storage = empty(shape=shape, dtype=dtype, backend=backend, aligned_index=aligned_index,
                    dimensions=dimensions, memory_class=memory_class)
    storage[...] = fill_value
    return storage


def ones(shape: Sequence[int], dtype=np.float64, *, backend: str,
         aligned_index: Optional[Sequence[int]] = None, dimensions: Optional[Sequence[str]] = None,
         memory_class: Optional[int] = None):
    storage = empty(shape=shape, dtype=dtype, backend=backend, aligned_index=aligned_index,
                    dimensions=dimensions, memory_class=memory_class)
    storage[...] = storage.dtype.type(1)
    return storage


def zeros(shape: Sequence[int], dtype=np.float64, *, backend: str,
          aligned_index: Optional[Sequence[int]] = None, dimensions: Optional[Sequence[str]] = None,
         memory_class: Optional[int] = None):
    storage = empty(shape=shape, dtype=dtype, backend=backend, aligned_index=aligned_index,
                    dimensions=dimensions, memory_class=memory_class)
    storage[...] = storage.dtype.type(0)
    return storage


def from_array(data, dtype=np.float64, *, backend: str, aligned_index: Optional[Sequence[int]] = None,
               dimensions: Optional[Sequence[str]] = None, memory_class: Optional[int] = None):
    """Copy ``data`` into a new optimally laid-out array (interface.py:264-327)."""
    host = asnumpy(data) if isinstance(data, DeviceArray) else np.asarray(data)
    shape = host.shape
    if dtype is None:
        dtype = host.dtype
    dtype = np.dtype(dtype)
    if dtype.shape:
        if not shape[-dtype.ndim:] == dtype.shape:
            raise ValueError(f"Incompatible data shape {shape} with dtype of shape {dtype.shape}.")
        shape = shape[: -dtype.ndim]
    storage = empty(shape=shape, dtype=dtype, backend=backend, aligned_index=aligned_index,
                    dimensions=dimensions, memory_class=memory_class)
    base = dtype.base if dtype.shape else dtype
    storage[...] = host.astype(base, copy=False)
    return storage
