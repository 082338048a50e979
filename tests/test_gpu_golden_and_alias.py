"""GPU: the HIP kernels against the committed golden vectors DIRECTLY (not through the oracle), and the alias
rules of the C ABI.

* tests/golden/stencils_small.npz holds inputs and outputs of the three stencils executed as the statement code of
  SURVEY.md Appendix A on the reference's own ``Field`` shim (scripts/make_golden.py).  GPU -> oracle -> golden is
  transitive; these tests close the triangle.
* The reference's numpy backend evaluates a right-hand side completely before it assigns
  (/root/reference/src/gt4py/cartesian/gtc/numpy/npir_codegen.py:205-210), so passing one array for two fields is
  well defined there.  The library either reproduces that result (aliases that do not depend on the evaluation
  order: `out` sharing its array with another field of the tridiagonal solve, `coeff` with `out_field`) or refuses
  the call -- compared here with the oracle called on the SAME aliased numpy arrays.
"""

import pathlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_numpy as R  # noqa: E402  (oracle = checker only)

GOLD = np.load(pathlib.Path(__file__).parent / "golden" / "stencils_small.npz")
LAYOUTS = ["ifirst", "ifirst_unaligned", "kfirst", "jfirst"]


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_laplacian_kernel_reproduces_the_golden_vector(layout):
    import gpu_util as G

    inp, out = G.DevArray(GOLD["lap_inp"], layout), G.DevArray(GOLD["lap_out0"], layout)
    G.lap5(inp, out, tuple(GOLD["lap_origin_inp"]), tuple(GOLD["lap_origin_out"]), tuple(GOLD["lap_domain"]), variant=0)
    assert _same(out.get(), GOLD["lap_out"])


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_hdiff_kernel_reproduces_the_golden_vector(tag, layout):
    import gpu_util as G
    from gt4py_amd import _lib

    o_in, o_cf, o_out = (tuple(int(v) for v in o) for o in GOLD[f"hd_{tag}_origins"])
    d_in, d_cf = G.DevArray(GOLD[f"hd_{tag}_in"], layout), G.DevArray(GOLD[f"hd_{tag}_coeff"], layout)
    d_out = G.DevArray(GOLD[f"hd_{tag}_out0"], layout)
    G.hdiff(d_in, d_out, d_cf, o_in, o_out, o_cf, tuple(GOLD[f"hd_{tag}_domain"]), _lib.HDIFF_LIMITER)
    assert _same(d_out.get(), GOLD[f"hd_{tag}_out"])


@pytest.mark.parametrize("layout", LAYOUTS)
def test_tridiagonal_kernel_reproduces_the_golden_vector(layout):
    import gpu_util as G

    o = tuple(int(v) for v in GOLD["tri_origin"])
    d = [G.DevArray(GOLD[n], layout) for n in ("tri_inf", "tri_diag", "tri_sup0", "tri_rhs0")]
    d.append(G.DevArray(np.zeros_like(GOLD["tri_out"]), layout))
    G.tridiag(*d, {n: o for n in ("inf", "diag", "sup", "rhs", "out")}, tuple(GOLD["tri_domain"]))
    assert _same(d[2].get(), GOLD["tri_sup"]) and _same(d[3].get(), GOLD["tri_rhs"]) and _same(d[4].get(), GOLD["tri_out"])


def test_golden_vectors_through_the_user_api():
    """storage -> @gtscript.stencil(backend="hip:mi300") -> __call__ with per-field origins -> golden outputs."""
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    backend = "hip:mi300"
    lap = gtscript.stencil(backend=backend, definition=hip_templates.lap_notebook, dtypes={"T": np.float64})
    inp = gt_storage.from_array(GOLD["lap_inp"], backend=backend)
    out = gt_storage.from_array(GOLD["lap_out0"], backend=backend)
    lap(inp, out, origin={"inp": tuple(GOLD["lap_origin_inp"]), "out": tuple(GOLD["lap_origin_out"])},
        domain=tuple(GOLD["lap_domain"]))
    assert _same(out.get(), GOLD["lap_out"])
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        hd = gtscript.stencil(backend=backend, definition=hip_templates.hdiff_limiter_field, dtypes={"T": dt})
        o_in, o_cf, o_out = (tuple(int(v) for v in o) for o in GOLD[f"hd_{tag}_origins"])
        f = {n: gt_storage.from_array(GOLD[f"hd_{tag}_{k}"], dt, backend=backend)
             for n, k in (("in_field", "in"), ("coeff", "coeff"), ("out_field", "out0"))}
        hd(**f, origin={"in_field": o_in, "coeff": o_cf, "out_field": o_out}, domain=tuple(GOLD[f"hd_{tag}_domain"]))
        assert _same(f["out_field"].get(), GOLD[f"hd_{tag}_out"])
    tri = gtscript.stencil(backend=backend, definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64})
    o = tuple(int(v) for v in GOLD["tri_origin"])
    f = {n: gt_storage.from_array(GOLD[k], backend=backend) for n, k in
         (("inf", "tri_inf"), ("diag", "tri_diag"), ("sup", "tri_sup0"), ("rhs", "tri_rhs0"))}
    f["out"] = gt_storage.zeros(GOLD["tri_out"].shape, backend=backend)
    tri(**f, origin=o, domain=tuple(GOLD["tri_domain"]))
    assert _same(f["out"].get(), GOLD["tri_out"]) and _same(f["sup"].get(), GOLD["tri_sup"]) and _same(f["rhs"].get(), GOLD["tri_rhs"])


# ---- aliases ---------------------------------------------------------------------------------------------
def _tridiag_inputs(shape, seed=7):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, shape), rng.uniform(4, 5, shape), rng.uniform(-1, 1, shape), rng.uniform(-10, 10, shape))


@pytest.mark.parametrize("layout", ["ifirst", "kfirst"])
@pytest.mark.parametrize("shape", [(5, 4, 9), (70, 3, 40), (66, 2, 100)])  # the last two reach the on-chip-stack kernels
@pytest.mark.parametrize("shared_with", ["rhs", "sup", "inf", "diag"])
def test_tridiagonal_out_may_share_its_array(shared_with, shape, layout):
    import gpu_util as G

    names = ("inf", "diag", "sup", "rhs")
    host = dict(zip(names, _tridiag_inputs(shape)))
    # oracle with the SAME aliasing: `out` is the very numpy array of `shared_with`
    want = {n: a.copy() for n, a in host.items()}
    R.tridiag(want["inf"], want["diag"], want["sup"], want["rhs"], want[shared_with])
    dev = {n: G.DevArray(a, layout) for n, a in host.items()}
    origins = {n: (0, 0, 0) for n in names + ("out",)}
    G.tridiag(dev["inf"], dev["diag"], dev["sup"], dev["rhs"], dev[shared_with], origins, shape)
    for n in names:
        assert _same(dev[n].get(), want[n]), f"field {n} with out sharing {shared_with}"


@pytest.mark.parametrize("pair", [("sup", "rhs"), ("sup", "diag"), ("rhs", "inf")])
def test_tridiagonal_refuses_aliases_that_depend_on_the_evaluation_order(pair):
    import gpu_util as G
    from gt4py_amd import _lib

    shape = (6, 5, 8)
    host = dict(zip(("inf", "diag", "sup", "rhs"), _tridiag_inputs(shape)))
    dev = {n: G.DevArray(a, "ifirst") for n, a in host.items()}
    dev["out"] = G.DevArray(np.zeros(shape), "ifirst")
    dev[pair[1]] = dev[pair[0]]
    before = {n: dev[n].get().copy() for n in dev}
    with pytest.raises(_lib.NativeError, match="overlap in memory"):
        G.tridiag(dev["inf"], dev["diag"], dev["sup"], dev["rhs"], dev["out"], {n: (0, 0, 0) for n in dev}, shape)
    assert all(np.array_equal(dev[n].get(), before[n]) for n in dev), "a refused call must not touch the fields"


def test_tridiagonal_refuses_a_shifted_view_of_the_same_buffer():
    import gpu_util as G
    from gt4py_amd import _lib

    shape = (6, 5, 12)
    inf, diag, sup, rhs = _tridiag_inputs(shape)
    dev = [G.DevArray(a, "ifirst") for a in (inf, diag, sup, rhs)]
    origins = {"inf": (0, 0, 0), "diag": (0, 0, 0), "sup": (0, 0, 0), "rhs": (0, 0, 0), "out": (0, 0, 2)}
    with pytest.raises(_lib.NativeError, match="overlap in memory"):  # out = rhs two levels up: not the same elements
        G.tridiag(dev[0], dev[1], dev[2], dev[3], dev[3], origins, (6, 5, 10))


@pytest.mark.parametrize("layout", ["ifirst", "kfirst"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hdiff_out_field_may_be_the_coefficient_array(dtype, layout):
    import gpu_util as G
    from gt4py_amd import _lib

    rng = np.random.default_rng(17)
    u = rng.uniform(-10, 10, (70, 21, 3)).astype(dtype)
    c = rng.uniform(0, 0.5, u.shape).astype(dtype)
    want = c.copy()
    R.hdiff(u, want, want)  # the oracle with out_field IS coeff
    d_u, d_c = G.DevArray(u, layout, (2, 2, 0)), G.DevArray(c, layout, (2, 2, 0))
    G.hdiff(d_u, d_c, d_c, (2, 2, 0), (2, 2, 0), (2, 2, 0), (66, 17, 3), _lib.HDIFF_LIMITER)
    assert _same(d_c.get(), want)


def test_in_place_horizontal_stencils_are_refused():
    """Every point reads its neighbours' OLD values in the reference; an in-place launch cannot provide that."""
    import gpu_util as G
    from gt4py_amd import _lib

    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (20, 18, 4))
    d_a = G.DevArray(a, "ifirst", (2, 2, 0))
    d_c = G.DevArray(rng.uniform(0, 0.5, a.shape), "ifirst", (2, 2, 0))
    with pytest.raises(_lib.NativeError, match="overlap in memory"):
        G.lap5(d_a, d_a, (1, 1, 0), (1, 1, 0), (18, 16, 4))
    with pytest.raises(_lib.NativeError, match="overlap in memory"):
        G.lap5(d_a, d_a, (1, 1, 0), (2, 1, 0), (17, 16, 4))  # shifted by one column: still overlapping
    with pytest.raises(_lib.NativeError, match="overlap in memory"):
        G.hdiff(d_a, d_a, d_c, (2, 2, 0), (2, 2, 0), (2, 2, 0), (16, 14, 4), _lib.HDIFF_LIMITER)
    with pytest.raises(_lib.NativeError, match="without being the same elements"):
        G.hdiff(d_a, d_c, d_c, (2, 2, 0), (2, 2, 0), (3, 2, 0), (15, 14, 4), _lib.HDIFF_LIMITER)
    assert np.array_equal(d_a.get(), a)
    # ... and through the user API the same refusal surfaces as an exception, never as a wrong answer
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64})
    f = gt_storage.from_array(a, backend="hip:mi300", aligned_index=(1, 1, 0))
    with pytest.raises(Exception, match="overlap in memory"):
        lap(f, f, origin=(1, 1, 0))
    assert np.array_equal(f.get(), a)


def test_disjoint_halves_of_one_buffer_are_not_an_alias():
    """Two fields inside one allocation (slices of a bigger array) are fine as long as their elements differ."""
    import torch

    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates

    rng = np.random.default_rng(5)
    host = rng.uniform(-1, 1, (2, 34, 30, 6))
    both = torch.from_numpy(host).cuda()
    from gt4py_amd.storage.device_array import DeviceArray

    inp, out = DeviceArray(both[0]), DeviceArray(both[1])
    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64})
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # K-contiguous slices: the layout warning is not the point here
        lap(inp, out, origin=(1, 1, 0))
    want = host[1].copy()
    R.laplacian(host[0], want)
    assert np.array_equal(out.get(), want)


def test_element_disjoint_views_of_one_buffer_run():
    """interleaved slices (`vel[..., 0]` / `vel[..., 1]`) and the J halves of an I-contiguous
    parent have overlapping BYTE ranges but share no element; the library proves that (common.hip.h:
    elements_disjoint, hip_generic.elements_disjoint) and runs them -- kernel library and generic executor."""
    import warnings

    import torch

    import oracle.numpy_backend  # noqa: F401
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_templates
    from gt4py_amd.storage.device_array import DeviceArray

    rng = np.random.default_rng(11)
    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64})
    gen = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64},
                           use_kernel_library=False)
    hd = gtscript.stencil(backend="hip:mi300", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float64})
    # 1. interleaved: the last axis of a (I, J, K, 2) array
    host = rng.uniform(-1, 1, (34, 30, 6, 2))
    for stencil in (lap, gen):
        both = torch.from_numpy(host).cuda()
        inp, out = DeviceArray(both[..., 0]), DeviceArray(both[..., 1])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            stencil(inp, out, origin=(1, 1, 0))
        want = host[..., 1].copy()
        R.laplacian(host[..., 0], want)
        assert np.array_equal(out.get(), want)
        assert np.array_equal(inp.get(), host[..., 0])
    # 2. J halves of one I-contiguous parent (strides (1, pitch, pitch * 2 nj)): the halves' byte ranges interleave per level
    ni, nj, nk = 40, 20, 5
    parent = torch.zeros((nk, 2 * nj, ni), dtype=torch.float64, device="cuda").permute(2, 1, 0)  # I contiguous
    u, c = rng.uniform(-10, 10, (ni, nj, nk)), rng.uniform(0, 0.5, (ni, nj, nk))
    lower, upper = parent[:, :nj, :], parent[:, nj:, :]
    lower.copy_(torch.from_numpy(u))
    d_c = DeviceArray(torch.from_numpy(c).cuda())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hd(DeviceArray(lower), DeviceArray(upper), d_c, origin=(2, 2, 0), domain=(ni - 4, nj - 4, nk))
    want = np.zeros_like(u)
    R.hdiff(u, want, c)
    assert np.array_equal(upper.cpu().numpy(), want)
    # 3. ... but a view shifted by one row inside the same half still overlaps and is refused
    with pytest.raises(Exception, match="overlap in memory"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lap(DeviceArray(parent[:, 0:10, :]), DeviceArray(parent[:, 1:11, :]), origin=(1, 1, 0), domain=(ni - 2, 8, nk))
