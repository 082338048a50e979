"""ctypes binding of ``libgt4py_amd.so`` (the C ABI declared in ``include/gt4py_amd.h``).

This is the Python side of the host->native boundary that the reference crosses with a
per-stencil pybind11 module (``pyext_module.run_computation(...)``,
/root/reference/src/gt4py/cartesian/backend/gtc_common.py:144-168).  There is no CPU fallback:
if the shared library is missing or does not load, every entry point raises ``RuntimeError``.
"""

from __future__ import annotations

import ctypes
import os
import pathlib
import threading
from typing import Optional, Sequence


_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libgt4py_amd.so"
HEADER_PATH = _HERE.parent / "include" / "gt4py_amd.h"

#: every symbol include/gt4py_amd.h declares (kept in sync by tests/test_c_abi.py)
EXPORTED_SYMBOLS = (
    "gt4mi_abi_version",
    "gt4mi_last_error",
    "gt4mi_device_info",
    "gt4mi_stream_sync",
    "gt4mi_lap5_f64",
    "gt4mi_lap5_f32",
    "gt4mi_hdiff_f64",
    "gt4mi_hdiff_f32",
    "gt4mi_hdiff_ring_f64",
    "gt4mi_hdiff_ring_f32",
    "gt4mi_lap5_ring_f64",
    "gt4mi_lap5_ring_f32",
    "gt4mi_tridiag_f64",
    "gt4mi_tridiag_f32",
    "gt4mi_halo_pack",
    "gt4mi_halo_unpack",
    "gt4mi_comm_unique_id",
    "gt4mi_comm_create",
    "gt4mi_comm_create_local",
    "gt4mi_comm_destroy",
    "gt4mi_comm_info",
    "gt4mi_halo_plan_create",
    "gt4mi_halo_plan_destroy",
    "gt4mi_halo_plan_set_option",
    "gt4mi_halo_plan_direct_prepare",
    "gt4mi_halo_plan_direct_layout",
    "gt4mi_halo_plan_direct_connect",
    "gt4mi_halo_plan_direct_status",
    "gt4mi_dist_lap5_query",
    "gt4mi_halo_plan_concurrent",
    "gt4mi_halo_exchange",
    "gt4mi_halo_exchange_begin",
    "gt4mi_halo_exchange_fork",
    "gt4mi_halo_exchange_end",
    "gt4mi_dist_lap5_f64",
    "gt4mi_dist_lap5_f32",
    "gt4mi_dist_lap5_f64_pipelined",
    "gt4mi_dist_lap5_f64_wide",
    "gt4mi_dist_lap5_f64_skewed",
    "gt4mi_dist_hdiff_f64",
    "gt4mi_dist_hdiff_f32",
    "gt4mi_rtc_compile",
    "gt4mi_rtc_free",
    "gt4mi_module_load",
    "gt4mi_module_unload",
    "gt4mi_module_function",
    "gt4mi_function_info",
    "gt4mi_launch",
    "gt4mi_launch_batch",
    "gt4mi_stream_copy",
    "gt4mi_memory_write_probe",
)

GT4MI_ABI_VERSION = 7

# gt4mi_status
OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_OUT_OF_BOUNDS = -2
ERR_UNSUPPORTED = -3
ERR_HIP = -4
ERR_TIMEOUT = -5

# lap5 variants / flags
LAP_NOTEBOOK, LAP_DOCS, LAP_SUITE, LAP_AVG = 0, 1, 2, 3
LAP_LITERAL_F32 = 1
# gt4mi_halo_plan_set_option
PLAN_SCHEDULE, PLAN_INTERIOR_WG_PER_CU, PLAN_DEFER_JOIN, PLAN_EDGE_COLUMNS, PLAN_TRANSPORT, PLAN_DIRECT_TIMEOUT_MS, PLAN_DIRECT_FENCED = 0, 1, 2, 3, 4, 5, 6
TRANSPORT_RCCL, TRANSPORT_DIRECT = 0, 1
SCHEDULE_JOIN, SCHEDULE_CHAIN, SCHEDULE_SWAP, SCHEDULE_SWAP_PACKED, SCHEDULE_INLINE = 0, 1, 2, 3, 4
# hdiff flags
HDIFF_LIMITER, HDIFF_INTERNAL_F32, HDIFF_COEFF_F32 = 1, 2, 4

_Int3 = ctypes.c_int64 * 3


class Field(ctypes.Structure):
    """``gt4mi_field``: device pointer + shape + byte strides + origin."""

    _fields_ = [
        ("data", ctypes.c_void_p),
        ("shape", _Int3),
        ("stride", _Int3),
        ("origin", _Int3),
    ]

    @classmethod
    def make(cls, ptr: int, shape: Sequence[int], strides: Sequence[int], origin: Sequence[int]) -> "Field":
        if not (len(shape) == len(strides) == len(origin) == 3):
            raise ValueError("gt4mi_field describes exactly three axes (I, J, K)")
        return cls(ctypes.c_void_p(ptr), _Int3(*map(int, shape)), _Int3(*map(int, strides)),
                   _Int3(*map(int, origin)))


class HaloMsg(ctypes.Structure):
    """``gt4mi_halo_msg``: one box sent to / received from ``peer`` in ``phase`` 0 (I faces) or 1 (J faces)."""

    _fields_ = [("peer", ctypes.c_int32), ("phase", ctypes.c_int32), ("lo", _Int3), ("extent", _Int3)]

    @classmethod
    def make(cls, peer: int, phase: int, lo: Sequence[int], extent: Sequence[int]) -> "HaloMsg":
        return cls(int(peer), int(phase), _Int3(*map(int, lo)), _Int3(*map(int, extent)))


class DirectInfo(ctypes.Structure):
    """``gt4mi_direct_info``: what a rank's peers need to reach the receive buffers and flag words of its plan (plain bytes)."""

    _fields_ = [("pool_handle", ctypes.c_char * 64), ("pool_bytes", ctypes.c_int64), ("flag_words", ctypes.c_int64),
                ("pid", ctypes.c_int32), ("device", ctypes.c_int32)]


class ExecInfo(ctypes.Structure):
    """``gt4mi_exec_info``: host timestamps of the native call (run_cpp_start/end_time) and the interval its kernels
    spent on the device (run_hip_start/end_time, from a hipEvent pair; both 0.0 when nothing was launched)."""

    _fields_ = [("run_cpp_start_time", ctypes.c_double), ("run_cpp_end_time", ctypes.c_double),
                ("run_hip_start_time", ctypes.c_double), ("run_hip_end_time", ctypes.c_double)]


class NativeError(RuntimeError):
    """A libgt4py_amd entry point returned a non-zero status."""

    def __init__(self, func: str, status: int, message: str):
        super().__init__(f"{func} failed with status {status}: {message}")
        self.status = status


_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None


def _declare(lib: ctypes.CDLL) -> None:
    P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    FP = ctypes.POINTER(Field)
    DOM = ctypes.POINTER(ctypes.c_int64)
    EI = ctypes.POINTER(ExecInfo)
    lib.gt4mi_abi_version.restype = I
    lib.gt4mi_abi_version.argtypes = []
    lib.gt4mi_last_error.restype = ctypes.c_char_p
    lib.gt4mi_last_error.argtypes = []
    lib.gt4mi_device_info.restype = I
    lib.gt4mi_device_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.gt4mi_stream_sync.restype = I
    lib.gt4mi_stream_sync.argtypes = [P]
    for name in ("gt4mi_lap5_f64", "gt4mi_lap5_f32"):
        f = getattr(lib, name)
        f.restype = I
        f.argtypes = [DOM, FP, FP, I, I, P, EI]
    for name in ("gt4mi_hdiff_f64", "gt4mi_hdiff_f32"):
        f = getattr(lib, name)
        f.restype = I
        f.argtypes = [DOM, FP, FP, FP, D, I, P, EI]
    W4 = ctypes.POINTER(ctypes.c_int)
    for name in ("gt4mi_hdiff_ring_f64", "gt4mi_hdiff_ring_f32"):
        f = getattr(lib, name)
        f.restype = I
        f.argtypes = [DOM, FP, FP, FP, D, I, W4, P, EI]
    for name in ("gt4mi_lap5_ring_f64", "gt4mi_lap5_ring_f32"):
        f = getattr(lib, name)
        f.restype = I
        f.argtypes = [DOM, FP, FP, I, I, W4, W4, P, EI]
    for name in ("gt4mi_tridiag_f64", "gt4mi_tridiag_f32"):
        f = getattr(lib, name)
        f.restype = I
        f.argtypes = [DOM, FP, FP, FP, FP, FP, P, EI]
    lib.gt4mi_halo_pack.restype = I
    lib.gt4mi_halo_pack.argtypes = [FP, DOM, DOM, P, I, P]
    lib.gt4mi_halo_unpack.restype = I
    lib.gt4mi_halo_unpack.argtypes = [FP, DOM, DOM, P, I, P]
    lib.gt4mi_stream_copy.restype = I
    lib.gt4mi_stream_copy.argtypes = [P, P, ctypes.c_size_t, P]
    lib.gt4mi_memory_write_probe.restype = I
    lib.gt4mi_memory_write_probe.argtypes = [P, P, ctypes.c_size_t, I, P, ctypes.POINTER(D)]
    MP = ctypes.POINTER(HaloMsg)
    PP = ctypes.POINTER(ctypes.c_void_p)
    lib.gt4mi_comm_unique_id.restype = I
    lib.gt4mi_comm_unique_id.argtypes = [P]
    lib.gt4mi_comm_create_local.restype = I
    lib.gt4mi_comm_create_local.argtypes = [I, I, ctypes.POINTER(P)]
    lib.gt4mi_comm_create.restype = I
    lib.gt4mi_comm_create.argtypes = [P, I, I, PP]
    lib.gt4mi_comm_destroy.restype = I
    lib.gt4mi_comm_destroy.argtypes = [P]
    IP = ctypes.POINTER(ctypes.c_int)
    lib.gt4mi_comm_info.restype = I
    lib.gt4mi_comm_info.argtypes = [P, IP, IP, IP]
    lib.gt4mi_halo_plan_create.restype = I
    lib.gt4mi_halo_plan_create.argtypes = [P, I, MP, I, MP, I, PP]
    lib.gt4mi_halo_plan_destroy.restype = I
    lib.gt4mi_halo_plan_destroy.argtypes = [P]
    lib.gt4mi_halo_plan_set_option.restype = I
    lib.gt4mi_halo_plan_set_option.argtypes = [P, I, I]
    lib.gt4mi_halo_plan_direct_prepare.restype = I
    lib.gt4mi_halo_plan_direct_prepare.argtypes = [P, ctypes.POINTER(DirectInfo)]
    lib.gt4mi_halo_plan_direct_layout.restype = I
    lib.gt4mi_halo_plan_direct_layout.argtypes = [P, I, I, I, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    lib.gt4mi_halo_plan_direct_connect.restype = I
    lib.gt4mi_halo_plan_direct_connect.argtypes = [P, I, I, I, ctypes.POINTER(DirectInfo), ctypes.c_int64, I]
    lib.gt4mi_dist_lap5_query.restype = I
    lib.gt4mi_dist_lap5_query.argtypes = [P, DOM, FP, FP, I, ctypes.POINTER(ctypes.c_int)]
    lib.gt4mi_halo_plan_direct_status.restype = I
    lib.gt4mi_halo_plan_direct_status.argtypes = [P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint)]
    lib.gt4mi_halo_plan_concurrent.restype = I
    lib.gt4mi_halo_plan_concurrent.argtypes = [P]
    lib.gt4mi_halo_exchange.restype = I
    lib.gt4mi_halo_exchange.argtypes = [P, FP, P]
    lib.gt4mi_halo_exchange_begin.restype = I
    lib.gt4mi_halo_exchange_begin.argtypes = [P, FP, P]
    lib.gt4mi_halo_exchange_end.restype = I
    lib.gt4mi_halo_exchange_end.argtypes = [P, P]
    lib.gt4mi_halo_exchange_fork.restype = I
    lib.gt4mi_halo_exchange_fork.argtypes = [P, P]
    lib.gt4mi_dist_lap5_f64.restype = I
    lib.gt4mi_dist_lap5_f64.argtypes = [P, DOM, FP, FP, I, I, P]
    lib.gt4mi_dist_lap5_f32.restype = I
    lib.gt4mi_dist_lap5_f32.argtypes = [P, DOM, FP, FP, I, I, I, P]
    lib.gt4mi_dist_lap5_f64_pipelined.restype = I
    lib.gt4mi_dist_lap5_f64_pipelined.argtypes = [P, DOM, FP, FP, I, I, P]
    lib.gt4mi_dist_lap5_f64_wide.restype = I
    lib.gt4mi_dist_lap5_f64_wide.argtypes = [P, DOM, FP, FP, I, I, I, I, P]
    lib.gt4mi_dist_lap5_f64_skewed.restype = I
    lib.gt4mi_dist_lap5_f64_skewed.argtypes = [P, DOM, FP, FP, I, I, I, P]
    for name in ("gt4mi_dist_hdiff_f64", "gt4mi_dist_hdiff_f32"):
        f = getattr(lib, name)
        f.restype = I
        f.argtypes = [P, DOM, FP, FP, FP, D, I, I, P]
    SZ = ctypes.c_size_t
    lib.gt4mi_rtc_compile.restype = I
    lib.gt4mi_rtc_compile.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), I, PP,
                                      ctypes.POINTER(SZ), ctypes.c_char_p, SZ]
    lib.gt4mi_rtc_free.restype = I
    lib.gt4mi_rtc_free.argtypes = [P]
    lib.gt4mi_module_load.restype = I
    lib.gt4mi_module_load.argtypes = [P, PP]
    lib.gt4mi_module_unload.restype = I
    lib.gt4mi_module_unload.argtypes = [P]
    lib.gt4mi_module_function.restype = I
    lib.gt4mi_module_function.argtypes = [P, ctypes.c_char_p, PP]
    lib.gt4mi_function_info.restype = I
    lib.gt4mi_function_info.argtypes = [P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    U3 = ctypes.POINTER(ctypes.c_uint32)
    lib.gt4mi_launch.restype = I
    lib.gt4mi_launch.argtypes = [P, U3, U3, P, SZ, P, EI]
    lib.gt4mi_launch_batch.restype = I
    lib.gt4mi_launch_batch.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), U3, U3, ctypes.POINTER(ctypes.c_void_p),
                                       SZ, P, EI]


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises ``RuntimeError`` when it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = pathlib.Path(os.environ.get("GT4PY_AMD_LIB", LIB_PATH))
        if not path.exists():
            raise RuntimeError(
                f"libgt4py_amd.so not found at {path}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gt4py_amd/csrc`. "
                "The hip:mi300 backend has no CPU fallback."
            )
        try:
            lib = ctypes.CDLL(str(path))
        except OSError as exc:  # pragma: no cover - depends on the machine
            raise RuntimeError(f"cannot load {path}: {exc}") from exc
        _declare(lib)
        if lib.gt4mi_abi_version() != GT4MI_ABI_VERSION:
            raise RuntimeError(
                f"{path} has ABI version {lib.gt4mi_abi_version()}, expected {GT4MI_ABI_VERSION}"
            )
        _lib = lib
    return _lib


def check(func: str, status: int) -> None:
    if status != OK:
        raise NativeError(func, status, load().gt4mi_last_error().decode("utf-8", "replace"))


def domain3(domain: Sequence[int]):
    return _Int3(*map(int, domain))


def int4(values: Sequence[int]):
    """``const int[4]`` argument: ring widths towards {low I, high I, low J, high J}."""
    return (ctypes.c_int * 4)(*map(int, values))


def rtc_compile(source: str, name: str = "gt4mi_stencil.hip", options: Sequence[str] = ()) -> bytes:
    """HIP source -> gfx950 code object (bytes) through hiprtc inside the library.  Needs no GPU."""
    lib = load()
    code, size = ctypes.c_void_p(), ctypes.c_size_t()
    log = ctypes.create_string_buffer(1 << 16)
    opts = (ctypes.c_char_p * max(len(options), 1))(*[o.encode() for o in options])
    rc = lib.gt4mi_rtc_compile(source.encode(), name.encode(), opts, len(options), ctypes.byref(code),
                               ctypes.byref(size), log, len(log))
    if rc != OK:
        raise NativeError("gt4mi_rtc_compile", rc,
                          lib.gt4mi_last_error().decode("utf-8", "replace") + "\n" + log.value.decode("utf-8", "replace"))
    try:
        return ctypes.string_at(code, size.value)
    finally:
        lib.gt4mi_rtc_free(code)


def device_info() -> str:
    buf = ctypes.create_string_buffer(256)
    check("gt4mi_device_info", load().gt4mi_device_info(buf, 256))
    return buf.value.decode()
