"""GPU parity tests: hand-written HIP kernels (through the C ABI) vs the CPU oracle.

Bar: bit-exact against the numpy restatement (oracle/ref_numpy.py), which follows the numpy
backend's evaluation order and dtype rules.  The north-star tolerance (1e-12 in fp64) is therefore
met with margin; the tests assert equality and report the max abs difference on failure.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R  # noqa: E402  (oracle = checker only)

DOMAINS = [(1, 1, 1), (3, 5, 2), (17, 33, 5), (64, 64, 8), (65, 63, 7), (130, 40, 3), (300, 37, 2)]
LAYOUTS = ["ifirst", "ifirst_unaligned", "kfirst", "jfirst"]


def _eq(got, want, what=""):
    if not np.array_equal(got, want, equal_nan=True):
        diff = np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64)))
        nbad = int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum())
        raise AssertionError(f"{what}: {nbad} mismatching elements, max abs diff {diff:.3e}")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("domain", DOMAINS)
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_lap5_f64_parity(domain, layout, variant):
    import gpu_util as G

    rng = np.random.default_rng(1337)
    shape = (domain[0] + 2, domain[1] + 2, domain[2])
    inp = rng.uniform(-1, 1, shape)
    out0 = rng.uniform(-1, 1, shape)  # halo of `out` must stay untouched
    want = out0.copy()
    R.laplacian(inp, want, origin_inp=(1, 1, 0), origin_out=(1, 1, 0), domain=domain,
                variant=["notebook", "docs", "suite", "avg"][variant])
    d_in = G.DevArray(inp, layout, align_index=(1, 1, 0))
    d_out = G.DevArray(out0, layout, align_index=(1, 1, 0))
    G.lap5(d_in, d_out, (1, 1, 0), (1, 1, 0), domain, variant)
    _eq(d_out.get(), want, f"lap5 f64 {domain} {layout} v{variant}")


@pytest.mark.parametrize("literal32", [False, True])
@pytest.mark.parametrize("domain", [(17, 33, 5), (64, 64, 8), (300, 37, 2), (513, 6, 2)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("align", [(1, 1, 0), (0, 0, 0), (2, 0, 0)])
def test_lap5_f32_parity(domain, variant, literal32, align):
    import gpu_util as G
    from gt4py_amd import _lib

    rng = np.random.default_rng(7)
    shape = (domain[0] + 2, domain[1] + 2, domain[2])
    inp = rng.uniform(-1, 1, shape).astype(np.float32)
    # oracle: float64-literal semantics == compute in float64 from float32 inputs where the
    # expression tree says so; emulate by following the same tree explicitly
    c = inp[1:-1, 1:-1]
    w, e, s, n = inp[:-2, 1:-1], inp[2:, 1:-1], inp[1:-1, :-2], inp[1:-1, 2:]
    W = np.float32 if literal32 else np.float64
    if variant == 0:
        r = ((((W(-4.0) * c.astype(W)) + w.astype(W)) + e.astype(W)) + s.astype(W)) + n.astype(W)
    elif variant == 1:
        r = (W(-4.0) * c.astype(W)) + (((e + w) + n) + s).astype(W)
    elif variant == 2:
        r = (W(4.0) * c.astype(W)) - (((e + w) + n) + s).astype(W)
    else:
        r = W(0.25) * (((n + s) + e) + w).astype(W)
    want = np.zeros(shape, np.float32)
    want[1:-1, 1:-1] = r.astype(np.float32)
    # (origins 4 or 8 bytes off a 16-byte boundary and widths that are no multiple of 4 run 8-byte lanes with masked edges)
    d_in = G.DevArray(inp, "ifirst", align_index=align)
    d_out = G.DevArray(np.zeros(shape, np.float32), "ifirst", align_index=align)
    G.lap5(d_in, d_out, (1, 1, 0), (1, 1, 0), domain, variant, _lib.LAP_LITERAL_F32 if literal32 else 0)
    _eq(d_out.get(), want, f"lap5 f32 {domain} v{variant} lit32={literal32} aligned_index {align}")


def test_lap5_known_answers():
    """Reference KATs: x^2+y^2 -> 4 (examples/lap_cartesian_vs_next.ipynb cells 5-9);
    avg of ones -> 1 (test_call_interface.py:221-285)."""
    import gpu_util as G

    nx = ny = 32
    inp = np.fromfunction(lambda x, y, z: x**2 + y**2, (nx, ny, 1))
    d_in = G.DevArray(inp, "ifirst", align_index=(1, 1, 0))
    d_out = G.DevArray(np.zeros_like(inp), "ifirst", align_index=(1, 1, 0))
    G.lap5(d_in, d_out, (1, 1, 0), (1, 1, 0), (nx - 2, ny - 2, 1), 0)
    out = d_out.get()
    assert (out[1:-1, 1:-1] == 4.0).all()
    assert (out[0] == 0).all() and (out[-1] == 0).all() and (out[:, 0] == 0).all() and (out[:, -1] == 0).all()

    ones = np.ones((22, 22, 10))
    d_in = G.DevArray(ones, "ifirst", align_index=(1, 1, 0))
    d_out = G.DevArray(np.zeros_like(ones), "ifirst", align_index=(1, 1, 0))
    G.lap5(d_in, d_out, (2, 2, 0), (2, 2, 0), (10, 10, 10), 3)
    out = d_out.get()
    assert (out[2:12, 2:12, :] == 1).all() and out.sum() == 1000


def test_lap5_bounds_are_checked():
    import gpu_util as G
    from gt4py_amd import _lib

    a = G.DevArray(np.ones((22, 22, 10)), "ifirst")
    b = G.DevArray(np.zeros((22, 22, 10)), "ifirst")
    with pytest.raises(_lib.NativeError) as ei:
        G.lap5(a, b, (2, 2, 0), (2, 2, 0), (20, 20, 10), 3)
    assert ei.value.status == _lib.ERR_OUT_OF_BOUNDS
    with pytest.raises(_lib.NativeError):
        G.lap5(a, b, (0, 1, 0), (0, 1, 0), (5, 5, 5), 0)


# ------------------------------------------------------------------------------------------------
def _hdiff_inputs(domain, dtype, seed=2024):
    rng = np.random.default_rng(seed)
    ni, nj, nk = domain[0] + 4, domain[1] + 4, domain[2]
    x = np.arange(ni)[:, None, None] / max(ni, 2)
    y = np.arange(nj)[None, :, None] / max(nj, 2)
    f = 5 + 8 * (2 + np.cos(np.pi * (x + 1.5 * y)) + np.sin(2 * np.pi * (x + 1.5 * y))) / 4
    f = f + 0.1 * rng.uniform(-1, 1, (ni, nj, nk))
    coeff = rng.uniform(0.0, 0.05, (ni, nj, nk))
    return f.astype(dtype), coeff.astype(dtype)


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("limiter", [True, False])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("domain", DOMAINS)
def test_hdiff_parity_field_coeff(domain, dtype, limiter, layout):
    import gpu_util as G
    from gt4py_amd import _lib

    inp, coeff = _hdiff_inputs(domain, dtype)
    out0 = np.full(inp.shape, -7.0, dtype)
    want = out0.copy()
    R.hdiff(inp, want, coeff, origin_in=(2, 2, 0), origin_out=(2, 2, 0), origin_coeff=(2, 2, 0),
            domain=domain, limiter=limiter)
    d_in = G.DevArray(inp, layout, align_index=(2, 2, 0))
    d_cf = G.DevArray(coeff, layout, align_index=(2, 2, 0))
    d_out = G.DevArray(out0, layout, align_index=(2, 2, 0))
    G.hdiff(d_in, d_out, d_cf, (2, 2, 0), (2, 2, 0), (2, 2, 0), domain, _lib.HDIFF_LIMITER if limiter else 0)
    _eq(d_out.get(), want, f"hdiff {np.dtype(dtype).name} {domain} {layout} limiter={limiter}")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dJ", [1, 3, 4, 5, 15, 16, 17, 33])
@pytest.mark.parametrize("dI", [32, 33, 123, 124, 125, 126, 247, 248, 249, 250, 497])
def test_hdiff_at_the_partition_points_of_the_shared_rows_kernel(dI, dJ, dtype):
    """hdiff_share.hip.h (round 6) cuts a domain into strips of 62 lanes x 16 bytes (124 float64 / 248 float32 columns) and workgroups
    of 4 waves x 4 rows: every width around a strip boundary, every height around a wave / workgroup boundary -- the last strip
    with one column, lanes that straddle the readable columns, waves with 1-3 rows, idle waves behind the barrier -- against the
    oracle, with the halo and everything outside the domain untouched; scalar and field coefficient, limiter on."""
    import gpu_util as G
    from gt4py_amd import _lib

    domain = (dI, dJ, 2)
    inp, coeff = _hdiff_inputs(domain, dtype, seed=dI * 100 + dJ)
    for cf_field in (True, False):
        out0 = np.full(inp.shape, -7.0, dtype)
        want = out0.copy()
        weight = coeff if cf_field else np.float64(0.03)
        R.hdiff(inp, want, weight, origin_in=(2, 2, 0), origin_out=(2, 2, 0), origin_coeff=(2, 2, 0), domain=domain, limiter=True)
        d_in = G.DevArray(inp, "ifirst", align_index=(2, 2, 0))
        d_cf = G.DevArray(coeff, "ifirst", align_index=(2, 2, 0)) if cf_field else float(weight)
        d_out = G.DevArray(out0, "ifirst", align_index=(2, 2, 0))
        G.hdiff(d_in, d_out, d_cf, (2, 2, 0), (2, 2, 0), (2, 2, 0) if cf_field else None, domain, _lib.HDIFF_LIMITER)
        _eq(d_out.get(), want, f"hdiff {np.dtype(dtype).name} {domain} coeff {'field' if cf_field else 'scalar'}")


@pytest.mark.parametrize("dtype,lit32,c32", [(np.float64, False, False), (np.float32, False, False),
                                             (np.float32, True, True), (np.float32, True, False),
                                             (np.float32, False, True)])
@pytest.mark.parametrize("domain", [(17, 33, 5), (130, 40, 3)])
def test_hdiff_parity_scalar_coeff_and_precisions(domain, dtype, lit32, c32):
    """weight as a scalar parameter (test_suites.py:205-220) and literal_float_precision=32."""
    import gpu_util as G
    from gt4py_amd import _lib

    inp, _ = _hdiff_inputs(domain, dtype, seed=11)
    weight = np.float32(0.31) if c32 else np.float64(0.31)
    want = np.zeros_like(inp)
    R.hdiff(inp, want, weight, domain=domain, limiter=True, literal_float_precision=32 if lit32 else 64)
    flags = _lib.HDIFF_LIMITER | (_lib.HDIFF_INTERNAL_F32 if lit32 else 0) | (_lib.HDIFF_COEFF_F32 if c32 else 0)
    d_in = G.DevArray(inp, "ifirst", align_index=(2, 2, 0))
    d_out = G.DevArray(np.zeros_like(inp), "ifirst", align_index=(2, 2, 0))
    G.hdiff(d_in, d_out, float(weight), (2, 2, 0), (2, 2, 0), None, domain, flags)
    _eq(d_out.get(), want, f"hdiff scalar {np.dtype(dtype).name} lit32={lit32} c32={c32}")


def test_hdiff_reference_validation_function():
    """No-limiter hdiff against the reference's own numpy validation (test_suites.py:222-230),
    on the reference's input ranges (u in [-10,10], weight in [0,0.5], halo 2, fp64)."""
    import gpu_util as G

    rng = np.random.default_rng(3)
    for domain in [(1, 1, 1), (5, 9, 3), (15, 15, 15)]:
        u = rng.uniform(-10, 10, (domain[0] + 4, domain[1] + 4, domain[2]))
        weight = float(rng.uniform(0, 0.5))
        want = R.hdiff_validation(u, weight)
        d_in = G.DevArray(u, "ifirst", align_index=(2, 2, 0))
        d_out = G.DevArray(np.zeros(domain), "ifirst")
        G.hdiff(d_in, d_out, weight, (2, 2, 0), (0, 0, 0), None, domain, 0)
        _eq(d_out.get(), want, f"hdiff validation {domain}")


def test_hdiff_plane_is_identity():
    """lap of an affine plane is exactly 0 -> all fluxes 0 -> out == in bit-exactly."""
    import gpu_util as G
    from gt4py_amd import _lib

    i, j, k = np.meshgrid(np.arange(36.0), np.arange(28.0), np.arange(3.0), indexing="ij")
    plane = 3.0 * i - 2.0 * j + 0.5 * k + 7.0
    d_in = G.DevArray(plane, "ifirst", align_index=(2, 2, 0))
    d_out = G.DevArray(np.zeros_like(plane), "ifirst", align_index=(2, 2, 0))
    G.hdiff(d_in, d_out, 0.4, (2, 2, 0), (2, 2, 0), None, (32, 24, 3), _lib.HDIFF_LIMITER)
    assert np.array_equal(d_out.get()[2:-2, 2:-2], plane[2:-2, 2:-2])


def test_hdiff_nan_and_inf_propagate_like_numpy():
    import gpu_util as G
    from gt4py_amd import _lib

    inp, coeff = _hdiff_inputs((20, 12, 2), np.float64)
    inp[7, 6, 0] = np.nan
    inp[12, 5, 1] = np.inf
    want = np.zeros_like(inp)
    R.hdiff(inp, want, coeff, domain=(20, 12, 2), limiter=True)
    d_in, d_cf = G.DevArray(inp, "ifirst", (2, 2, 0)), G.DevArray(coeff, "ifirst", (2, 2, 0))
    d_out = G.DevArray(np.zeros_like(inp), "ifirst", (2, 2, 0))
    G.hdiff(d_in, d_out, d_cf, (2, 2, 0), (2, 2, 0), (2, 2, 0), (20, 12, 2), _lib.HDIFF_LIMITER)
    _eq(d_out.get(), want, "hdiff nan/inf")


# ------------------------------------------------------------------------------------------------
def _tridiag_inputs(shape, dtype, seed=7):
    rng = np.random.default_rng(seed)
    diag = rng.uniform(4, 5, shape).astype(dtype)
    inf = rng.uniform(-1, 1, shape).astype(dtype)
    sup = rng.uniform(-1, 1, shape).astype(dtype)
    rhs = rng.uniform(-10, 10, shape).astype(dtype)
    return inf, diag, sup, rhs


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(1, 1, 2), (3, 5, 2), (17, 33, 5), (64, 64, 8), (65, 63, 7), (40, 9, 160), (514, 3, 19),
                                   (34, 5, 50), (130, 3, 73), (70, 2, 33), (66, 4, 72),  # K around the on-chip stack sizes
                                   (66, 3, 56), (66, 3, 57), (70, 2, 58), (65, 2, 60), (130, 2, 64), (64, 2, 65), (66, 2, 71),
                                   (66, 3, 120), (66, 3, 121), (70, 2, 122), (65, 2, 128), (64, 2, 129), (130, 2, 135),
                                   (66, 2, 144), (66, 2, 145), (70, 2, 146), (65, 2, 147), (64, 2, 148), (130, 2, 149), (64, 3, 161)])
def test_tridiag_parity(shape, dtype, layout):
    import gpu_util as G

    inf, diag, sup, rhs = _tridiag_inputs(shape, dtype)
    s_w, r_w, o_w = sup.copy(), rhs.copy(), np.zeros(shape, dtype)
    R.tridiag(inf, diag, s_w, r_w, o_w)
    d = [G.DevArray(a, layout) for a in (inf, diag, sup, rhs, np.zeros(shape, dtype))]
    origins = {n: (0, 0, 0) for n in ("inf", "diag", "sup", "rhs", "out")}
    G.tridiag(*d, origins, shape)
    _eq(d[4].get(), o_w, f"tridiag out {shape} {layout}")
    _eq(d[2].get(), s_w, "tridiag sup (in-place)")
    _eq(d[3].get(), r_w, "tridiag rhs (in-place)")


def test_tridiag_against_scipy_and_residual():
    import gpu_util as G
    from scipy.linalg import solve_banded

    shape = (6, 5, 40)
    inf, diag, sup, rhs = _tridiag_inputs(shape, np.float64, seed=99)
    d = [G.DevArray(a, "ifirst") for a in (inf, diag, sup, rhs, np.zeros(shape))]
    G.tridiag(*d, {n: (0, 0, 0) for n in ("inf", "diag", "sup", "rhs", "out")}, shape)
    x = d[4].get()
    for i in range(shape[0]):
        for j in range(shape[1]):
            ab = np.zeros((3, shape[2]))
            ab[0, 1:] = sup[i, j, :-1]
            ab[1] = diag[i, j]
            ab[2, :-1] = inf[i, j, 1:]
            ref = solve_banded((1, 1), ab, rhs[i, j])
            assert np.abs(ref - x[i, j]).max() <= 1e-12
    res = diag * x - rhs
    res[:, :, 1:] += inf[:, :, 1:] * x[:, :, :-1]
    res[:, :, :-1] += sup[:, :, :-1] * x[:, :, 1:]
    assert np.abs(res).max() <= 1e-13 * 10 * 5


def test_tridiag_subdomain_with_origins_and_singular_pivot():
    """origin/domain smaller than the arrays; untouched outside; 0/0 propagates as in numpy (N3)."""
    import gpu_util as G

    shape = (9, 8, 12)
    inf, diag, sup, rhs = _tridiag_inputs(shape, np.float64, seed=5)
    diag[3, 3, 0] = 0.0
    rhs[3, 3, 0] = 0.0
    origins = {"inf": (1, 2, 0), "diag": (1, 2, 0), "sup": (1, 2, 1), "rhs": (1, 2, 0), "out": (0, 0, 2)}
    domain = (7, 5, 10)
    out0 = np.full(shape, 3.0)
    s_w, r_w, o_w = sup.copy(), rhs.copy(), out0.copy()
    R.tridiag(inf, diag, s_w, r_w, o_w, origins=origins, domain=domain)
    d = [G.DevArray(a, "ifirst") for a in (inf, diag, sup, rhs, out0)]
    G.tridiag(*d, origins, domain)
    _eq(d[4].get(), o_w, "tridiag out sub-domain")
    _eq(d[2].get(), s_w, "tridiag sup sub-domain")
    _eq(d[3].get(), r_w, "tridiag rhs sub-domain")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_halo_pack_unpack_roundtrip(dtype):
    import ctypes

    import gpu_util as G
    import torch
    from gt4py_amd import _lib

    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, (20, 14, 6)).astype(dtype)
    d = G.DevArray(a, "ifirst", (2, 2, 0))
    lo, ext = (3, 1, 0), (2, 11, 6)
    buf = torch.empty(int(np.prod(ext)), dtype=G.TORCH_DT[np.dtype(dtype)], device="cuda")
    G.call("gt4mi_halo_pack", ctypes.byref(d.field((0, 0, 0))), _lib.domain3(lo), _lib.domain3(ext),
           buf.data_ptr(), a.dtype.itemsize, G.stream_ptr())
    got = buf.cpu().numpy().reshape(ext[::-1]).transpose(2, 1, 0)
    assert np.array_equal(got, a[3:5, 1:12, 0:6])
    # unpack into another place of a second array
    b = np.zeros_like(a)
    db = G.DevArray(b, "kfirst")
    G.call("gt4mi_halo_unpack", ctypes.byref(db.field((0, 0, 0))), _lib.domain3((10, 2, 0)), _lib.domain3(ext),
           buf.data_ptr(), a.dtype.itemsize, G.stream_ptr())
    want = b.copy()
    want[10:12, 2:13, :] = a[3:5, 1:12, :]
    assert np.array_equal(db.get(), want)


def test_stream_copy():
    import gpu_util as G
    import torch

    src = torch.arange(1 << 20, dtype=torch.float64, device="cuda")
    dst = torch.zeros_like(src)
    G.call("gt4mi_stream_copy", src.data_ptr(), dst.data_ptr(), src.numel() * 8, G.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(src, dst)


@pytest.mark.parametrize("domain", [(0, 6, 4), (6, 0, 4), (6, 6, 0)])
def test_empty_domains_are_no_ops_at_the_boundary(domain):
    """One empty axis: every entry point returns GT4MI_OK without a launch and without a write (the numpy backend's
    slices are empty, stencil_object.py:370-373 only refuses the all-zero domain).  The tridiagonal entry point keeps
    its own contract for K (min_sequential_axis_size 2)."""
    import gpu_util as G
    from gt4py_amd import _lib

    rng = np.random.default_rng(3)
    a = rng.standard_normal((10, 10, 6))
    d_in, d_out, d_cf = G.DevArray(a, "ifirst", (2, 2, 0)), G.DevArray(np.full_like(a, 7.0), "ifirst", (2, 2, 0)), \
        G.DevArray(a * 0.1, "ifirst", (2, 2, 0))
    G.lap5(d_in, d_out, (2, 2, 0), (2, 2, 0), domain, 0)
    G.hdiff(d_in, d_out, d_cf, (2, 2, 0), (2, 2, 0), (2, 2, 0), domain, _lib.HDIFF_LIMITER)
    G.hdiff(d_in, d_out, 0.25, (2, 2, 0), (2, 2, 0), None, domain, 0)
    assert (d_out.get() == 7.0).all() and (d_in.get() == a).all()
    if domain[2] >= 2:
        d = [G.DevArray(np.full((10, 10, 6), float(n + 1)), "ifirst") for n in range(5)]
        G.tridiag(*d, {n: (0, 0, 0) for n in ("inf", "diag", "sup", "rhs", "out")}, domain)
        for n, arr in enumerate(d):
            assert (arr.get() == float(n + 1)).all()


# ---- boundary-ring kernels (csrc/hdiff_ring.hip.h, lap5_ring.hip.h): one launch for the four boxes of a ring ---------
def _ring_mask(shape, origin, domain, outer, inner):
    """Boolean mask of (domain grown by outer[W, E, S, N]) minus (domain shrunk by inner[...]) in array indices."""
    m = np.zeros(shape[:2], dtype=bool)
    oi, oj = origin[0], origin[1]
    di, dj = domain[0], domain[1]
    m[oi - outer[0]:oi + di + outer[1], oj - outer[2]:oj + dj + outer[3]] = True
    m[oi + inner[0]:oi + di - inner[1], oj + inner[2]:oj + dj - inner[3]] = False
    return m


@pytest.mark.parametrize("layout", ["ifirst", "ifirst_unaligned", "kfirst"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("widths", [(2, 2, 2, 2), (0, 2, 2, 0), (2, 0, 0, 0), (0, 0, 0, 2), (2, 2, 0, 0), (1, 0, 2, 3), (32, 32, 2, 2),
                                    (16, 0, 2, 0), (6, 3, 2, 2)])
@pytest.mark.parametrize("domain", [(130, 70, 3), (7, 9, 2), (64, 5, 4)])
def test_hdiff_ring_equals_the_whole_domain_kernel_on_the_ring(domain, widths, dtype, layout):
    """gt4mi_hdiff_ring_*: the points within widths[side] of a side get exactly the values of the oracle (and of the
    whole-domain kernel); everything else keeps what the output array held.  W / E boxes exactly 2 columns wide run the
    transposed tile, other widths (1, 3, 6, 16, 32: the fused distributed step asks for 32) J-march strips."""
    import gpu_util as G
    from gt4py_amd import _lib

    if widths[0] + widths[1] > domain[0] or widths[2] + widths[3] > domain[1]:
        pytest.skip("ring wider than the domain")
    rng = np.random.default_rng(hash((domain, widths)) % 2**32)
    shape = (domain[0] + 4, domain[1] + 4, domain[2])
    u = rng.uniform(-10, 10, shape).astype(dtype)
    c = rng.uniform(0, 0.5, shape).astype(dtype)
    want_full = np.zeros_like(u)
    R.hdiff(u, want_full, c)
    for flags, coeff in ((_lib.HDIFF_LIMITER, "field"), (0, 0.125)):
        if coeff != "field":  # a scalar weight declared with the fields' dtype, no limiter
            weight = np.float32(coeff) if dtype == np.float32 else np.float64(coeff)
            want_full = np.zeros_like(u)
            R.hdiff(u, want_full, weight, limiter=False)
        sentinel = np.full(shape, -777.0, dtype=dtype)
        d_u, d_o = G.DevArray(u, layout, (2, 2, 0)), G.DevArray(sentinel, layout, (2, 2, 0))
        d_c = G.DevArray(c, layout, (2, 2, 0)) if coeff == "field" else coeff
        G.hdiff_ring(d_u, d_o, d_c, (2, 2, 0), (2, 2, 0), (2, 2, 0), domain,
                     flags | (_lib.HDIFF_COEFF_F32 if coeff != "field" and dtype == np.float32 else 0), widths)
        mask = _ring_mask(shape, (2, 2, 0), domain, (0, 0, 0, 0), widths)
        want = np.where(mask[:, :, None], want_full, sentinel)
        got = d_o.get()
        assert np.array_equal(got, want), (flags, np.argwhere(got != want)[:5])


@pytest.mark.parametrize("layout", ["ifirst", "ifirst_unaligned", "jfirst"])
@pytest.mark.parametrize("outer,inner", [((0, 0, 0, 0), (1, 1, 1, 1)), ((0, 0, 0, 0), (0, 0, 1, 1)), ((0, 0, 1, 1), (0, 0, 3, 3)),
                                         ((1, 1, 1, 1), (3, 3, 3, 3)), ((2, 0, 0, 2), (4, 0, 0, 4)), ((0, 0, 0, 0), (2, 2, 2, 2)),
                                         ((0, 3, 0, 0), (0, 7, 0, 0)), ((0, 0, 3, 0), (0, 0, 7, 0)), ((0, 0, 0, 0), (16, 16, 1, 1)),
                                         ((0, 0, 0, 0), (8, 0, 1, 0)), ((0, 0, 0, 0), (0, 16, 0, 0)), ((0, 0, 0, 0), (20, 0, 0, 0))])
@pytest.mark.parametrize("domain", [(128, 40, 3), (70, 33, 2), (520, 18, 2)])
@pytest.mark.parametrize("variant", [0, 3])
def test_lap5_ring_equals_the_whole_domain_kernel_on_the_ring(domain, outer, inner, layout, variant):
    """gt4mi_lap5_ring_f64: (domain grown by outer) minus (domain shrunk by inner) gets the oracle's values -- also in the
    ghost region, where the time-skewed stepper computes redundantly -- and nothing else is written."""
    import gpu_util as G

    H = 4
    rng = np.random.default_rng(hash((domain, outer, inner)) % 2**32)
    shape = (domain[0] + 2 * H, domain[1] + 2 * H, domain[2])
    u = rng.uniform(-1, 1, shape)
    want_full = np.zeros_like(u)
    R.laplacian(u, want_full, variant=["notebook", "docs", "suite", "avg"][variant])  # on [1, -1) of the whole array
    sentinel = np.full(shape, -777.0)
    d_u, d_o = G.DevArray(u, layout, (H, H, 0)), G.DevArray(sentinel, layout, (H, H, 0))
    G.lap5_ring(d_u, d_o, (H, H, 0), (H, H, 0), domain, outer, inner, variant=variant)
    mask = _ring_mask(shape, (H, H, 0), domain, outer, inner)
    want = np.where(mask[:, :, None], want_full, sentinel)
    got = d_o.get()
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]


@pytest.mark.parametrize("domain", [(512, 9, 3), (511, 6, 2), (513, 5, 2), (1030, 5, 2), (126, 20, 4), (640, 4, 2)])
@pytest.mark.parametrize("variant", [0, 2])
@pytest.mark.parametrize("align", [(0, 0, 0), (1, 1, 0)])
def test_lap5_origin_off_the_aligned_column_and_odd_widths(domain, variant, align):
    """A compute-domain origin one column past a 16-byte boundary (storage allocated with the default aligned_index, origin
    (1, 1, 0)) and odd widths run the 16-byte-lane strips with masked edges (lap5_strip_lane<..., MASKED>; five-wave
    workgroups where the extra lane would otherwise need a second one) -- same values as the oracle, halo of `out` untouched."""
    import gpu_util as G

    rng = np.random.default_rng(sum(domain) + variant)
    shape = (domain[0] + 2, domain[1] + 2, domain[2])
    inp = rng.uniform(-1, 1, shape)
    out0 = rng.uniform(-1, 1, shape)
    want = out0.copy()
    R.laplacian(inp, want, domain=domain, variant=["notebook", "docs", "suite", "avg"][variant])
    d_in, d_out = G.DevArray(inp, "ifirst", align_index=align), G.DevArray(out0, "ifirst", align_index=align)
    G.lap5(d_in, d_out, (1, 1, 0), (1, 1, 0), domain, variant)
    _eq(d_out.get(), want, f"lap5 f64 {domain} aligned_index {align} v{variant}")


@pytest.mark.parametrize("align", [(0, 0, 0), (1, 1, 0), (3, 0, 0), (2, 2, 0)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("domain", [(130, 20, 3), (257, 9, 2), (61, 33, 2), (4, 4, 1), (124, 5, 2), (125, 16, 1), (248, 17, 2), (249, 4, 1)])
def test_hdiff_origin_off_the_aligned_column(domain, dtype, align):
    """Horizontal diffusion from an origin that lies 1 - 3 items past a 16-byte boundary (default aligned_index with origin
    (2, 2, 0) puts float32 fields 8 bytes off): the J-march strips keep their 16-byte lanes, started `lead` columns further
    left (hdiff_jmarch_strip), whole domain and boundary ring alike; same values as the oracle, nothing else written."""
    import gpu_util as G
    from gt4py_amd import _lib

    rng = np.random.default_rng(sum(domain))
    shape = (domain[0] + 4, domain[1] + 4, domain[2])
    u = rng.uniform(-10, 10, shape).astype(dtype)
    c = rng.uniform(0, 0.5, shape).astype(dtype)
    want = np.full(shape, -5.0, dtype=dtype)
    R.hdiff(u, want, c)
    d_u, d_c = G.DevArray(u, "ifirst", align), G.DevArray(c, "ifirst", align)
    d_o = G.DevArray(np.full(shape, -5.0, dtype=dtype), "ifirst", align)
    G.hdiff(d_u, d_o, d_c, (2, 2, 0), (2, 2, 0), (2, 2, 0), domain, _lib.HDIFF_LIMITER)
    _eq(d_o.get(), want, f"hdiff {np.dtype(dtype).name} {domain} aligned_index {align}")
    if domain[0] >= 40 and domain[1] >= 8:  # the ring with the widths the fused distributed step uses
        widths = (16, 16, 2, 2)
        d_r = G.DevArray(np.full(shape, -5.0, dtype=dtype), "ifirst", align)
        G.hdiff_ring(d_u, d_r, d_c, (2, 2, 0), (2, 2, 0), (2, 2, 0), domain, _lib.HDIFF_LIMITER, widths)
        mask = _ring_mask(shape, (2, 2, 0), domain, (0, 0, 0, 0), widths)
        _eq(d_r.get(), np.where(mask[:, :, None], want, dtype(-5.0)), f"hdiff ring {np.dtype(dtype).name} {domain} aligned_index {align}")
