"""The C-ABI shared library: loads, exports every symbol include/gt4py_amd.h declares, reports errors.

No compute call is made here (no GPU needed); argument validation that happens before any HIP call
is exercised because it is part of the boundary contract (negative status + thread-local message).
"""

import ctypes
import pathlib
import re

import pytest

from gt4py_amd import _lib

ROOT = pathlib.Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "gt4py_amd.h"


def _declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gt4mi_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    assert _lib.LIB_PATH.exists(), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/gt4py_amd.h but not exported"
    assert _lib.load().gt4mi_abi_version() == _lib.GT4MI_ABI_VERSION


def test_header_cites_the_reference_interface_it_replaces():
    text = HEADER.read_text()
    for needle in ("gtc_common.py:65-103", "gtcpp_backend.py:77-106", "stencil_definitions.py:316-328",
                   "stencil_definitions.py:219-232", "lap_cartesian_vs_next.ipynb"):
        assert needle in text


def test_argument_errors_are_reported_without_touching_the_gpu():
    lib = _lib.load()
    dom = _lib.domain3((4, 4, 4))
    # null field
    rc = lib.gt4mi_lap5_f64(dom, None, None, 0, 0, None, None)
    assert rc == _lib.ERR_INVALID_ARGUMENT and b"null" in lib.gt4mi_last_error()
    # halo does not fit: shape 4 cannot hold origin 1 + domain 4 + halo 1
    f = _lib.Field.make(0x1000, (4, 4, 4), (8, 32, 128), (1, 1, 0))
    rc = lib.gt4mi_lap5_f64(dom, ctypes.byref(f), ctypes.byref(f), 0, 0, None, None)
    assert rc == _lib.ERR_OUT_OF_BOUNDS and b"too small" in lib.gt4mi_last_error()
    # origin smaller than the stencil's reach
    g = _lib.Field.make(0x1000, (8, 8, 4), (8, 64, 512), (0, 1, 0))
    rc = lib.gt4mi_lap5_f64(dom, ctypes.byref(g), ctypes.byref(g), 0, 0, None, None)
    assert rc == _lib.ERR_OUT_OF_BOUNDS and b"origin" in lib.gt4mi_last_error()
    # bad variant
    h = _lib.Field.make(0x1000, (8, 8, 4), (8, 64, 512), (1, 1, 0))
    h_out = _lib.Field.make(0x100000, (8, 8, 4), (8, 64, 512), (1, 1, 0))
    rc = lib.gt4mi_lap5_f64(dom, ctypes.byref(h), ctypes.byref(h_out), 99, 0, None, None)
    assert rc == _lib.ERR_INVALID_ARGUMENT
    # output overlapping the input: refused before anything is launched (the reference evaluates the right-hand
    # side before it assigns, npir_codegen.py:205-210; an in-place kernel cannot)
    rc = lib.gt4mi_lap5_f64(dom, ctypes.byref(h), ctypes.byref(h), 0, 0, None, None)
    assert rc == _lib.ERR_UNSUPPORTED and b"overlap in memory" in lib.gt4mi_last_error()
    shifted = _lib.Field.make(0x1000 + 8 * 64 * 2, (8, 8, 4), (8, 64, 512), (1, 1, 0))
    rc = lib.gt4mi_lap5_f64(dom, ctypes.byref(h), ctypes.byref(shifted), 0, 0, None, None)
    assert rc == _lib.ERR_UNSUPPORTED and b"overlap in memory" in lib.gt4mi_last_error()
    # tridiagonal needs at least two levels (min_sequential_axis_size)
    t = _lib.Field.make(0x1000, (4, 4, 1), (8, 32, 128), (0, 0, 0))
    refs = [ctypes.byref(t)] * 5
    rc = lib.gt4mi_tridiag_f64(_lib.domain3((4, 4, 1)), *refs, None, None)
    assert rc == _lib.ERR_INVALID_ARGUMENT and b"at least 2" in lib.gt4mi_last_error()
    with pytest.raises(_lib.NativeError) as ei:
        _lib.check("gt4mi_tridiag_f64", rc)
    assert ei.value.status == _lib.ERR_INVALID_ARGUMENT
    # stream_copy alignment contract
    assert lib.gt4mi_stream_copy(0x1001, 0x2000, 32, None) == _lib.ERR_INVALID_ARGUMENT


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("GT4PY_AMD_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_no_hot_kernel_of_the_library_spills():
    """The build keeps the compiler's per-kernel resource remarks (csrc/Makefile -> lib/libgt4py_amd.resources.log).
    The column kernels keep up to 104 levels of two fields in registers: one level too many, or the `#pragma unroll`ed
    level loop left rolled (LLVM's default unroll budget), and the register arrays silently become scratch memory --
    a 30 % slower kernel with identical results, which no parity test would notice."""
    log = _lib.LIB_PATH.with_name("libgt4py_amd.resources.log")
    assert log.exists(), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    text = log.read_text()
    kernels = re.findall(r"remark: Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", text, re.S)
    by_name = {name: (int(scratch), int(waves)) for name, scratch, waves in kernels}
    hot = {n: v for n, v in by_name.items() if any(k in n for k in ("lap5_strip_kernel", "hdiff_jmarch_kernel", "tridiag_pipe_kernel",
                                                                   "tridiag_kernel", "halo_copy_kernel", "lap5_step_kernel", "lap5_edge_kernel"))}
    assert len(hot) >= 20, sorted(by_name)[:5]
    spilling = {n: v for n, v in hot.items() if v[0] != 0}
    assert not spilling, spilling
    # the deep tridiagonal variants are built for one wave per SIMD (all 512 registers of a lane)
    deep = [v for n, v in hot.items() if "tridiag_pipe_kernelIdLi104ELi40ELi4E" in n or "tridiag_pipe_kernelIdLi80ELi40ELi8E" in n]
    # (four: 104 + 40 with two waves per workgroup and nontemporal loads -- the default --, with one wave (GT4MI_TRIDIAG_WPB=1), with one
    # wave and plain loads (GT4MI_TRIDIAG_NT_LOADS=0) for A/B runs; 80 + 40)
    assert len(deep) == 4 and all(w == 1 for _, w in deep)
