"""Pin the CPU oracle: golden vectors, the reference's known-answer tests, independent restatements.

The oracle (oracle/ref_numpy.py, oracle/cpu_ifirst.c, oracle/numpy_backend.py) is what the GPU
parity tests compare against, so it is checked here against
  * tests/golden/stencils_small.npz -- Appendix-A statement code run on the reference's own
    ``Field`` shim (scripts/make_golden.py),
  * the reference's known answers (lap(x^2+y^2) = 4, avg(ones) = 1, hdiff validation function),
  * point-by-point loop restatements and scipy.linalg.solve_banded.
"""

import pathlib

import numpy as np
import pytest

from oracle import ref_numpy as R

GOLD = np.load(pathlib.Path(__file__).parent / "golden" / "stencils_small.npz")


def test_laplacian_golden_vector():
    out = GOLD["lap_out0"].copy()
    R.laplacian(GOLD["lap_inp"], out, origin_inp=tuple(GOLD["lap_origin_inp"]),
                origin_out=tuple(GOLD["lap_origin_out"]), domain=tuple(GOLD["lap_domain"]))
    assert np.array_equal(out, GOLD["lap_out"])


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_hdiff_golden_vector(tag):
    o_in, o_cf, o_out = (tuple(o) for o in GOLD[f"hd_{tag}_origins"])
    out = GOLD[f"hd_{tag}_out0"].copy()
    R.hdiff(GOLD[f"hd_{tag}_in"], out, GOLD[f"hd_{tag}_coeff"], origin_in=o_in, origin_out=o_out,
            origin_coeff=o_cf, domain=tuple(GOLD[f"hd_{tag}_domain"]), limiter=True)
    assert out.dtype == GOLD[f"hd_{tag}_out"].dtype
    assert np.array_equal(out, GOLD[f"hd_{tag}_out"])


def test_tridiag_golden_vector():
    sup, rhs, out = GOLD["tri_sup0"].copy(), GOLD["tri_rhs0"].copy(), np.zeros_like(GOLD["tri_out"])
    o = tuple(GOLD["tri_origin"])
    R.tridiag(GOLD["tri_inf"], GOLD["tri_diag"], sup, rhs, out,
              origins={n: o for n in ("inf", "diag", "sup", "rhs", "out")}, domain=tuple(GOLD["tri_domain"]))
    assert np.array_equal(out, GOLD["tri_out"])
    assert np.array_equal(sup, GOLD["tri_sup"])
    assert np.array_equal(rhs, GOLD["tri_rhs"])


def test_laplacian_known_answers():
    # examples/lap_cartesian_vs_next.ipynb cells 5-9: inp = x^2 + y^2 -> 4 in the interior, 0 border
    inp = np.fromfunction(lambda x, y, z: x**2 + y**2, (32, 32, 1))
    out = np.zeros_like(inp)
    R.laplacian(inp, out, domain=(30, 30, 1))
    assert (out[1:-1, 1:-1] == 4.0).all() and out.sum() == 4.0 * 900
    # test_suites.py:214 form gives -4; test_call_interface.py avg of ones gives 1
    R.laplacian(inp, out, domain=(30, 30, 1), variant="suite")
    assert (out[1:-1, 1:-1] == -4.0).all()
    ones, o = np.ones((22, 22, 10)), np.zeros((22, 22, 10))
    R.laplacian(ones, o, origin_inp=(2, 2, 0), origin_out=(2, 2, 0), domain=(10, 10, 10), variant="avg")
    assert (o[2:12, 2:12] == 1).all() and o.sum() == 1000


@pytest.mark.parametrize("variant", ["notebook", "docs"])
def test_laplacian_variants_agree_to_rounding(variant):
    rng = np.random.default_rng(0)
    inp = rng.uniform(-1, 1, (12, 11, 3))
    a, b = np.zeros_like(inp), np.zeros_like(inp)
    R.laplacian(inp, a, variant=variant)
    R.laplacian(inp, b, variant="suite")
    np.testing.assert_allclose(a, -b, rtol=0, atol=1e-14)


def test_laplacian_matches_loop_restatement():
    rng = np.random.default_rng(1337)
    inp = rng.uniform(-1, 1, (9, 8, 4))
    a, b = np.zeros((8, 7, 5)), np.zeros((8, 7, 5))
    kw = dict(origin_inp=(1, 1, 1), origin_out=(2, 0, 2), domain=(6, 5, 3))
    R.laplacian(inp, a, **kw)
    R.laplacian_loops(inp, b, **kw)
    assert np.array_equal(a, b)


def test_hdiff_no_limiter_equals_reference_validation():
    """test_suites.py:222-230 on the reference's input ranges."""
    rng = np.random.default_rng(42)
    for domain in [(1, 1, 1), (4, 7, 3), (15, 15, 15)]:
        u = rng.uniform(-10, 10, (domain[0] + 4, domain[1] + 4, domain[2]))
        w = float(rng.uniform(0, 0.5))
        out = np.zeros(domain)
        R.hdiff(u, out, w, origin_in=(2, 2, 0), origin_out=(0, 0, 0), domain=domain, limiter=False)
        assert np.array_equal(out, R.hdiff_validation(u, w))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("limiter", [True, False])
def test_hdiff_matches_loop_restatement(dtype, limiter):
    rng = np.random.default_rng(5)
    u = rng.uniform(-10, 10, (13, 11, 3)).astype(dtype)
    c = rng.uniform(0, 0.5, (13, 11, 3)).astype(dtype)
    a, b = np.zeros_like(u), np.zeros_like(u)
    R.hdiff(u, a, c, limiter=limiter)
    R.hdiff_loops(u, b, c, origin_in=(2, 2, 0), origin_out=(2, 2, 0), origin_coeff=(2, 2, 0),
                  domain=(9, 7, 3), limiter=limiter)
    assert np.array_equal(a, b)
    # the limiter really fires on a non-trivial subset of this input
    if limiter:
        n = np.zeros_like(u)
        R.hdiff(u, n, c, limiter=False)
        frac = (n[2:-2, 2:-2] != a[2:-2, 2:-2]).mean()
        assert frac > 0.05


def test_hdiff_invariants():
    # affine plane: lap == 0 exactly -> identity
    i, j, k = np.meshgrid(np.arange(12.0), np.arange(10.0), np.arange(2.0), indexing="ij")
    plane = 2.0 * i - 3.0 * j + k + 1.0
    out = np.zeros_like(plane)
    R.hdiff(plane, out, 0.3, limiter=True)
    assert np.array_equal(out[2:-2, 2:-2], plane[2:-2, 2:-2])
    # x^2+y^2: lap == -4 everywhere -> all fluxes 0 -> identity
    quad = i**2 + j**2
    R.hdiff(quad, out, 0.3, limiter=True)
    assert np.array_equal(out[2:-2, 2:-2], quad[2:-2, 2:-2])


def test_hdiff_f32_computes_in_f64_by_default():
    """with float64 literals the float32 stencil differs from an all-float32 one."""
    rng = np.random.default_rng(9)
    u = rng.uniform(-10, 10, (40, 40, 2)).astype(np.float32)
    c = rng.uniform(0, 0.5, (40, 40, 2)).astype(np.float32)
    a, b = np.zeros_like(u), np.zeros_like(u)
    R.hdiff(u, a, c, literal_float_precision=64)
    R.hdiff(u, b, c, literal_float_precision=32)
    assert a.dtype == b.dtype == np.float32
    assert not np.array_equal(a, b)
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-4)


def test_tridiag_matches_loops_and_scipy():
    from scipy.linalg import solve_banded

    rng = np.random.default_rng(7)
    shape = (5, 4, 23)
    inf, diag = rng.uniform(-1, 1, shape), rng.uniform(4, 5, shape)
    sup, rhs = rng.uniform(-1, 1, shape), rng.uniform(-10, 10, shape)
    s1, r1, o1 = sup.copy(), rhs.copy(), np.zeros(shape)
    s2, r2, o2 = sup.copy(), rhs.copy(), np.zeros(shape)
    R.tridiag(inf, diag, s1, r1, o1)
    R.tridiag_loops(inf, diag, s2, r2, o2)
    assert np.array_equal(o1, o2) and np.array_equal(s1, s2) and np.array_equal(r1, r2)
    for i in range(shape[0]):
        for j in range(shape[1]):
            ab = np.zeros((3, shape[2]))
            ab[0, 1:] = sup[i, j, :-1]
            ab[1] = diag[i, j]
            ab[2, :-1] = inf[i, j, 1:]
            assert np.abs(solve_banded((1, 1), ab, rhs[i, j]) - o1[i, j]).max() <= 1e-12
    res = diag * o1 - rhs
    res[:, :, 1:] += inf[:, :, 1:] * o1[:, :, :-1]
    res[:, :, :-1] += sup[:, :, :-1] * o1[:, :, 1:]
    assert np.abs(res).max() <= 1e-13 * 50


def test_tridiag_trivial_cases():
    # inf = sup = 0 -> out == rhs / diag with a single rounding
    rng = np.random.default_rng(3)
    shape = (3, 2, 6)
    diag, rhs = rng.uniform(1, 2, shape), rng.uniform(-1, 1, shape)
    z = np.zeros(shape)
    s, r, o = z.copy(), rhs.copy(), z.copy()
    R.tridiag(z, diag, s, r, o)
    assert np.array_equal(o, rhs / diag)
    # all ones divides 0/0 at k = 1 exactly like the reference smoke test (N3): nan, no exception
    ones = np.ones(shape)
    s, r, o = ones.copy(), ones.copy(), z.copy()
    R.tridiag(ones, ones, s, r, o)
    assert np.isnan(o).any() or np.isinf(o).any()


# ---- C restatement (oracle/cpu_ifirst.c) ------------------------------------------------------------
def test_c_restatement_matches_numpy_restatement():
    from oracle import cpu_ifirst as C

    if not C.available():
        pytest.skip("oracle/_build/libcpu_ifirst.so not built (run __graft_entry__.build())")
    rng = np.random.default_rng(11)
    # Laplacian, I-contiguous storage
    inp = np.asfortranarray(rng.uniform(-1, 1, (35, 22, 6)))
    want, got = np.zeros_like(inp), np.asfortranarray(np.zeros_like(inp))
    R.laplacian(inp, want)
    C.lap5_f64(inp, got, (1, 1, 0), (1, 1, 0), (33, 20, 6), threads=2)
    assert np.array_equal(got, want)
    # hdiff f64 / f32
    for dt in (np.float64, np.float32):
        u = np.asfortranarray(rng.uniform(-10, 10, (29, 31, 4)).astype(dt))
        c = np.asfortranarray(rng.uniform(0, 0.5, (29, 31, 4)).astype(dt))
        want, got = np.zeros_like(u), np.asfortranarray(np.zeros_like(u))
        R.hdiff(u, want, c)
        C.hdiff(u, got, c, (2, 2, 0), (2, 2, 0), (2, 2, 0), (25, 27, 4), threads=2)
        assert np.array_equal(got, want)
    # tridiagonal
    shape = (17, 9, 31)
    inf, diag = (np.asfortranarray(rng.uniform(a, b, shape)) for a, b in ((-1, 1), (4, 5)))
    sup, rhs = (np.asfortranarray(rng.uniform(a, b, shape)) for a, b in ((-1, 1), (-10, 10)))
    s1, r1, o1 = sup.copy(), rhs.copy(), np.zeros(shape)
    s2, r2, o2 = (np.asfortranarray(x.copy()) for x in (sup, rhs, np.zeros(shape)))
    R.tridiag(inf, diag, s1, r1, o1)
    C.tridiag_f64(inf, diag, s2, r2, o2, shape, threads=2)
    assert np.array_equal(o2, o1) and np.array_equal(s2, s1) and np.array_equal(r2, r1)


def test_c_restatement_is_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """Sanitizers belong on the CPU build (GPU AddressSanitizer is not available on this pool): oracle/selftest_cpu_ifirst.c
    runs every entry point of oracle/cpu_ifirst.c on heap arrays exactly as large as the stencils' reach requires, on both
    layouts and on degenerate domains, under -fsanitize=address,undefined with recovery off."""
    import pathlib
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    src = pathlib.Path(__file__).resolve().parent.parent / "oracle" / "selftest_cpu_ifirst.c"
    exe = tmp_path / "selftest_cpu_ifirst"
    build = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                            "-ffp-contract=off", "-Wall", "-o", str(exe), str(src), "-lm"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "clean under the sanitizers" in run.stdout, (run.stdout + run.stderr)[-2000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
