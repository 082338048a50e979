"""Helpers for the GPU parity tests: device buffers with a chosen memory layout + C-ABI calls."""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from gt4py_amd import _lib

TORCH_DT = {np.dtype("float64"): torch.float64, np.dtype("float32"): torch.float32}


class DevArray:
    """A device copy of a numpy IJK array with a selectable layout.

    layout "ifirst"  : I contiguous, rows padded so that element [align_index[0], j, k] is aligned to
                       ``align_bytes`` (what gt4py storages for gt:gpu / hip:mi300 look like)
    layout "ifirst_unaligned": I contiguous, rows padded to an odd pitch and base offset by one item
    layout "kfirst"  : C order of (I, J, K) -- K contiguous (numpy backend default)
    layout "jfirst"  : J contiguous
    """

    def __init__(self, host: np.ndarray, layout: str = "ifirst", align_index=(0, 0, 0), align_bytes=256):
        assert host.ndim == 3
        self.host_shape = host.shape
        self.dtype = host.dtype
        isz = host.dtype.itemsize
        ni, nj, nk = host.shape
        tdt = TORCH_DT[host.dtype]
        if layout == "ifirst":
            items = align_bytes // isz
            pitch = -(-ni // items) * items
            lead = (items - align_index[0] % items) % items
            # over-allocate; find an aligned base inside
            flat = torch.empty(pitch * nj * nk + 2 * items, dtype=tdt, device="cuda")
            base_off = (-(flat.data_ptr() // isz) % items + lead) % items
            self.strides = (1, pitch, pitch * nj)
            self._flat = flat
            self.offset = base_off
        elif layout == "ifirst_unaligned":
            pitch = ni + 3 if (ni + 3) % 2 else ni + 4
            flat = torch.empty(pitch * nj * nk + 8, dtype=tdt, device="cuda")
            self.strides = (1, pitch, pitch * nj)
            self._flat = flat
            self.offset = 1 if (flat.data_ptr() // isz) % 2 == 0 else 2
        elif layout == "kfirst":
            flat = torch.empty(ni * nj * nk, dtype=tdt, device="cuda")
            self.strides = (nj * nk, nk, 1)
            self._flat = flat
            self.offset = 0
        elif layout == "jfirst":
            flat = torch.empty(ni * nj * nk, dtype=tdt, device="cuda")
            self.strides = (nj, 1, ni * nj)
            self._flat = flat
            self.offset = 0
        else:
            raise ValueError(layout)
        self._flat.fill_(float("nan"))
        self.view = torch.as_strided(self._flat, host.shape, self.strides, self.offset)
        self.view.copy_(torch.from_numpy(np.ascontiguousarray(host)))

    @property
    def ptr(self) -> int:
        return self._flat.data_ptr() + self.offset * self.dtype.itemsize

    def field(self, origin) -> _lib.Field:
        bs = tuple(s * self.dtype.itemsize for s in self.strides)
        return _lib.Field.make(self.ptr, self.host_shape, bs, origin)

    def get(self) -> np.ndarray:
        torch.cuda.synchronize()
        return self.view.cpu().numpy()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args) -> None:
    lib = _lib.load()
    _lib.check(name, getattr(lib, name)(*args))


def lap5(inp: DevArray, out: DevArray, origin_in, origin_out, domain, variant=0, flags=0):
    name = "gt4mi_lap5_f64" if inp.dtype == np.float64 else "gt4mi_lap5_f32"
    call(name, _lib.domain3(domain), ctypes.byref(inp.field(origin_in)), ctypes.byref(out.field(origin_out)),
         variant, flags, stream_ptr(), None)


def hdiff(inp: DevArray, out: DevArray, coeff, origin_in, origin_out, origin_coeff, domain, flags):
    name = "gt4mi_hdiff_f64" if inp.dtype == np.float64 else "gt4mi_hdiff_f32"
    if isinstance(coeff, DevArray):
        cf, cs = ctypes.byref(coeff.field(origin_coeff)), 0.0
    else:
        cf, cs = None, float(coeff)
    call(name, _lib.domain3(domain), ctypes.byref(inp.field(origin_in)), ctypes.byref(out.field(origin_out)),
         cf, cs, flags, stream_ptr(), None)


def tridiag(inf, diag, sup, rhs, out, origins, domain):
    name = "gt4mi_tridiag_f64" if inf.dtype == np.float64 else "gt4mi_tridiag_f32"
    fs = [ctypes.byref(a.field(origins[n])) for n, a in
          zip(("inf", "diag", "sup", "rhs", "out"), (inf, diag, sup, rhs, out))]
    call(name, _lib.domain3(domain), *fs, stream_ptr(), None)


def hdiff_ring(inp: DevArray, out: DevArray, coeff, origin_in, origin_out, origin_coeff, domain, flags, widths):
    name = "gt4mi_hdiff_ring_f64" if inp.dtype == np.float64 else "gt4mi_hdiff_ring_f32"
    if isinstance(coeff, DevArray):
        cf, cs = ctypes.byref(coeff.field(origin_coeff)), 0.0
    else:
        cf, cs = None, float(coeff)
    call(name, _lib.domain3(domain), ctypes.byref(inp.field(origin_in)), ctypes.byref(out.field(origin_out)),
         cf, cs, flags, _lib.int4(widths), stream_ptr(), None)


def lap5_ring(inp: DevArray, out: DevArray, origin_in, origin_out, domain, outer, inner, variant=0, flags=0):
    name = "gt4mi_lap5_ring_f64" if inp.dtype == np.float64 else "gt4mi_lap5_ring_f32"
    call(name, _lib.domain3(domain), ctypes.byref(inp.field(origin_in)), ctypes.byref(out.field(origin_out)),
         variant, flags, _lib.int4(outer), _lib.int4(inner), stream_ptr(), None)
