"""``DeviceArray``: the ndarray-like object ``gt4py_amd.storage`` returns for GPU presets.

The reference returns a bare ``cupy.ndarray`` (on ROCm wrapped so that it exposes
``__cuda_array_interface__`` / ``__hip_array_interface__`` --
/root/reference/src/gt4py/storage/cartesian/utils.py:281-316).  cupy is not part of this stack;
device memory comes from PyTorch-ROCm's caching allocator (plumbing only), and this thin class gives
it the part of the ndarray interface stencil users touch: shape/strides/dtype, basic indexing and
assignment, host transfer, and the array-interface protocols for interop.
"""

from __future__ import annotations

import numbers
from typing import Any, Optional, Sequence, Tuple

import numpy as np

try:  # torch is plumbing for device memory; importing it must not be a hard failure on CPU-only use
    import torch
except Exception:  # pragma: no cover
    torch = None  # type: ignore


_NP_TO_TORCH = {}
if torch is not None:
    _NP_TO_TORCH = {
        np.dtype("float64"): torch.float64,
        np.dtype("float32"): torch.float32,
        np.dtype("float16"): torch.float16,
        np.dtype("int64"): torch.int64,
        np.dtype("int32"): torch.int32,
        np.dtype("int16"): torch.int16,
        np.dtype("int8"): torch.int8,
        np.dtype("uint8"): torch.uint8,
        np.dtype("bool"): torch.bool,
    }
    _TORCH_TO_NP = {v: k for k, v in _NP_TO_TORCH.items()}


def torch_dtype(dtype) -> "torch.dtype":
    try:
        return _NP_TO_TORCH[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"dtype {dtype} is not supported on the device") from None


def numpy_dtype(tdtype) -> np.dtype:
    return _TORCH_TO_NP[tdtype]


def _unwrap(value):
    return value._t if isinstance(value, DeviceArray) else value


class DeviceArray:
    """Strided view of HBM memory with numpy-style metadata.

    ``strides`` are in BYTES (numpy convention), ``dtype`` is a ``numpy.dtype``.
    """

    __slots__ = ("_t", "_owner", "__weakref__")
    __array_priority__ = 100

    def __init__(self, tensor, owner: Any = None):
        if torch is None or not isinstance(tensor, torch.Tensor):
            raise TypeError("DeviceArray wraps a torch.Tensor")
        self._t = tensor
        self._owner = owner  # keeps the raw allocation alive (the view already does; explicit)

    # ---- metadata ------------------------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, ...]:
        return tuple(self._t.shape)

    @property
    def ndim(self) -> int:
        return self._t.dim()

    @property
    def dtype(self) -> np.dtype:
        return numpy_dtype(self._t.dtype)

    @property
    def itemsize(self) -> int:
        return self._t.element_size()

    @property
    def strides(self) -> Tuple[int, ...]:
        isz = self._t.element_size()
        return tuple(s * isz for s in self._t.stride())

    @property
    def size(self) -> int:
        return self._t.numel()

    @property
    def nbytes(self) -> int:
        return self._t.numel() * self._t.element_size()

    @property
    def ptr(self) -> int:
        """Device address of element [0, ..., 0]."""
        return self._t.data_ptr()

    @property
    def device(self):
        return self._t.device

    @property
    def tensor(self):
        """The underlying torch view (shares memory)."""
        return self._t

    def __len__(self) -> int:
        return self.shape[0]

    def __repr__(self) -> str:
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, strides={self.strides}, device='{self.device}')"

    # ---- interop protocols ---------------------------------------------------------------
    @property
    def __cuda_array_interface__(self) -> dict:
        # same content as the reference's ROCm wrapper (storage/cartesian/utils.py:291-303)
        return {
            "shape": self.shape,
            "typestr": self.dtype.str,
            "descr": self.dtype.descr,
            "stream": 1,
            "version": 3,
            "strides": self.strides,
            "data": (self.ptr, False),
        }

    @property
    def __hip_array_interface__(self) -> dict:
        return self.__cuda_array_interface__

    def __dlpack__(self, stream=None):
        return self._t.__dlpack__() if stream is None else self._t.__dlpack__(stream=stream)

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()

    def __array__(self, dtype=None, copy=None):
        """Explicit device-to-host copy (``np.asarray(device_array)``)."""
        host = self.get()
        return host if dtype is None else host.astype(dtype)

    # ---- data movement -------------------------------------------------------------------
    def get(self) -> np.ndarray:
        """Copy to a new host ndarray (like ``cupy.ndarray.get``)."""
        return self._t.detach().cpu().numpy()

    def copy(self) -> "DeviceArray":
        """A copy with the SAME strides and the same alignment of its first element (modulo 512 bytes).

        ``torch.Tensor.clone`` keeps the strides only of dense tensors; a storage with padded rows would come back
        K-contiguous, i.e. in a layout the kernels take their any-stride path for (and a float32 field could lose the
        16-byte alignment of its origin column).  Padding elements between the rows are not copied."""
        t = self._t
        if t.numel() == 0 or t.dim() == 0:
            return DeviceArray(t.clone())
        isz = t.element_size()
        span = sum((n - 1) * s for n, s in zip(t.shape, t.stride())) + 1  # elements from the first to the last one
        raw = torch.empty(span * isz + 512, dtype=torch.uint8, device=t.device)
        offset = (t.data_ptr() - raw.data_ptr()) % 512
        flat = raw[offset: offset + span * isz].view(t.dtype)
        new = torch.as_strided(flat, tuple(t.shape), tuple(t.stride()))
        new.copy_(t)
        return DeviceArray(new, owner=raw)

    def fill(self, value) -> None:
        self._t.fill_(value)

    def transpose(self, *axes) -> "DeviceArray":
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if not axes:
            axes = tuple(reversed(range(self.ndim)))
        return DeviceArray(self._t.permute(*[int(a) for a in axes]), self._owner)

    # ---- indexing ------------------------------------------------------------------------
    @staticmethod
    def _key(key):
        if isinstance(key, tuple):
            return tuple(DeviceArray._key(k) for k in key)
        if isinstance(key, DeviceArray):
            return key._t
        if isinstance(key, numbers.Integral):
            return int(key)
        return key

    def __getitem__(self, key) -> "DeviceArray":
        return DeviceArray(self._t[self._key(key)], self._owner)

    def __setitem__(self, key, value) -> None:
        value = _unwrap(value)
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value)).to(self._t.device)
        elif isinstance(value, np.generic):
            value = value.item()
        elif not isinstance(value, (numbers.Number, torch.Tensor)) and hasattr(value, "__array__"):
            value = torch.from_numpy(np.ascontiguousarray(np.asarray(value))).to(self._t.device)
        self._t[self._key(key)] = value

    # ---- reductions / comparisons (enough for assertions in user tests) -------------------
    def _binary(self, other, op):
        other = _unwrap(other)
        if isinstance(other, np.ndarray):
            other = torch.from_numpy(np.ascontiguousarray(other)).to(self._t.device)
        return DeviceArray(op(self._t, other))

    def __eq__(self, other):  # type: ignore[override]
        return self._binary(other, lambda a, b: a == b)

    def __ne__(self, other):  # type: ignore[override]
        return self._binary(other, lambda a, b: a != b)

    def __lt__(self, other):
        return self._binary(other, lambda a, b: a < b)

    def __le__(self, other):
        return self._binary(other, lambda a, b: a <= b)

    def __gt__(self, other):
        return self._binary(other, lambda a, b: a > b)

    def __ge__(self, other):
        return self._binary(other, lambda a, b: a >= b)

    def __add__(self, other):
        return self._binary(other, lambda a, b: a + b)

    def __sub__(self, other):
        return self._binary(other, lambda a, b: a - b)

    def __mul__(self, other):
        return self._binary(other, lambda a, b: a * b)

    def __truediv__(self, other):
        return self._binary(other, lambda a, b: a / b)

    __hash__ = None  # type: ignore[assignment]

    def __bool__(self) -> bool:
        if self._t.numel() != 1:
            raise ValueError("The truth value of an array with more than one element is ambiguous.")
        return bool(self._t.item())

    def __float__(self) -> float:
        return float(self._t.item())

    def __int__(self) -> int:
        return int(self._t.item())

    def item(self):
        return self._t.item()

    def all(self) -> bool:
        return bool(self._t.all().item())

    def any(self) -> bool:
        return bool(self._t.any().item())

    def sum(self):
        return self._t.sum().item()

    def min(self):
        return self._t.min().item()

    def max(self):
        return self._t.max().item()


def asnumpy(array) -> np.ndarray:
    """Host copy of a device or host array (``cupy.asnumpy`` counterpart)."""
    if isinstance(array, DeviceArray):
        return array.get()
    if torch is not None and isinstance(array, torch.Tensor):
        return array.detach().cpu().numpy()
    return np.array(array)


def as_device_array(obj) -> DeviceArray:
    """Zero-copy view of any device buffer the reference's GPU path accepts.

    Accepted (SURVEY.md section 8b "Array protocol at the boundary"): DeviceArray, torch CUDA/ROCm
    tensors, objects exposing ``__cuda_array_interface__`` / ``__hip_array_interface__``, and
    DLPack exporters on a ROCm/CUDA device.  Host arrays are rejected: the reference would silently
    copy them to the device and drop the results (``cp.asarray``, storage/cartesian/utils.py:187-189).
    """
    if isinstance(obj, DeviceArray):
        return obj
    if hasattr(obj, "ndarray") and not hasattr(obj, "__cuda_array_interface__"):
        obj = obj.ndarray  # gt4py.next Field (utils.py:179-182)
        if isinstance(obj, DeviceArray):
            return obj
    if torch is None:
        raise RuntimeError("PyTorch-ROCm is required for device arrays")
    if isinstance(obj, torch.Tensor):
        if not obj.is_cuda:
            raise TypeError("a host torch.Tensor was passed to a GPU backend; move it to the device first")
        return DeviceArray(obj)
    if hasattr(obj, "__cuda_array_interface__") or hasattr(obj, "__hip_array_interface__"):
        return DeviceArray(torch.as_tensor(obj, device="cuda"), owner=obj)
    if hasattr(obj, "__dlpack_device__"):
        kind, _ = obj.__dlpack_device__()
        if int(kind) in (2, 10):  # kDLCUDA, kDLROCM (/root/reference/src/gt4py/_core/definitions.py:386-402)
            return DeviceArray(torch.from_dlpack(obj), owner=obj)
    raise TypeError(
        f"Cannot use {type(obj).__name__} as a device array for a GPU backend: expected a "
        "gt4py_amd.storage allocation, a torch ROCm tensor or a __cuda_array_interface__/DLPack exporter"
    )
