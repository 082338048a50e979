"""ORACLE (test infrastructure): ctypes wrapper + build recipe for oracle/cpu_ifirst.c.

``build()`` compiles the C restatement into ``oracle/_build/libcpu_ifirst.so`` (git-ignored; it
travels to the GPU box with the snapshot).  Only tests/, ``__graft_entry__`` and bench.py's
cpu_baseline leg use this module.
"""

from __future__ import annotations

import ctypes
import os
import pathlib
import subprocess
from typing import Optional, Sequence

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
SRC = HERE / "cpu_ifirst.c"
BUILD_DIR = HERE / "_build"
LIB = BUILD_DIR / "libcpu_ifirst.so"

_lib: Optional[ctypes.CDLL] = None


def build(march: str = "x86-64-v3", out: Optional[pathlib.Path] = None, quiet: bool = True) -> pathlib.Path:
    """gcc -O3 -fopenmp -ffp-contract=off (no FMA contraction: bit parity with numpy)."""
    out = pathlib.Path(out) if out is not None else LIB
    out.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["gcc", "-O3", f"-march={march}", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-shared",
           "-fPIC", "-o", str(out), str(SRC)]
    subprocess.run(cmd, check=True, capture_output=quiet)
    return out


def available() -> bool:
    return LIB.exists()


def load(path: Optional[pathlib.Path] = None) -> ctypes.CDLL:
    global _lib
    if path is None and _lib is not None:
        return _lib
    lib = ctypes.CDLL(str(path or LIB))
    i64, dp, fp = ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)
    lib.oracle_set_threads.argtypes = [ctypes.c_int]
    lib.oracle_max_threads.restype = ctypes.c_int
    lib.oracle_lap5_f64.argtypes = [dp, i64, i64, i64, dp, i64, i64, i64, i64, i64, i64]
    lib.oracle_hdiff_f64.argtypes = [dp, i64, i64, i64, dp, i64, i64, i64, dp, i64, i64, i64, i64, i64, i64, ctypes.c_int]
    lib.oracle_hdiff_f32.argtypes = [fp, i64, i64, i64, fp, i64, i64, i64, fp, i64, i64, i64, i64, i64, i64, ctypes.c_int]
    lib.oracle_tridiag_f64.argtypes = [dp, dp, dp, dp, dp, i64, i64, i64, i64, i64, i64]
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray, origin: Sequence[int]):
    isz = a.dtype.itemsize
    assert all(s % isz == 0 for s in a.strides)
    off = sum(int(o) * s for o, s in zip(origin, a.strides))
    ctype = ctypes.c_double if a.dtype == np.float64 else ctypes.c_float
    return ctypes.cast(a.ctypes.data + off, ctypes.POINTER(ctype)), [s // isz for s in a.strides]


def lap5_f64(inp, out, origin_inp, origin_out, domain, threads: int = 0, lib=None) -> None:
    lib = lib or load()
    if threads:
        lib.oracle_set_threads(threads)
    pi, si = _ptr(inp, origin_inp)
    po, so = _ptr(out, origin_out)
    lib.oracle_lap5_f64(pi, *si, po, *so, *map(int, domain))


def hdiff(inp, out, coeff, origin_in, origin_out, origin_coeff, domain, limiter: bool = True, threads: int = 0,
          lib=None) -> None:
    lib = lib or load()
    if threads:
        lib.oracle_set_threads(threads)
    pi, si = _ptr(inp, origin_in)
    po, so = _ptr(out, origin_out)
    pc, sc = _ptr(coeff, origin_coeff)
    fn = lib.oracle_hdiff_f64 if inp.dtype == np.float64 else lib.oracle_hdiff_f32
    fn(pi, *si, po, *so, pc, *sc, *map(int, domain), int(limiter))


def tridiag_f64(inf, diag, sup, rhs, out, domain, threads: int = 0, lib=None) -> None:
    lib = lib or load()
    if threads:
        lib.oracle_set_threads(threads)
    strides = {tuple(a.strides) for a in (inf, diag, sup, rhs, out)}
    assert len(strides) == 1, "tridiag oracle expects identically laid out fields"
    ptrs = [_ptr(a, (0, 0, 0)) for a in (inf, diag, sup, rhs, out)]
    lib.oracle_tridiag_f64(*[p for p, _ in ptrs], *ptrs[0][1], *map(int, domain))
