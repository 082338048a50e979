"""GPU: the full user path -- gt4py_amd.storage + @gtscript.stencil(backend="hip:mi300") -- vs the oracle.

These read like the reference's own integration tests (test_call_interface.py, test_suites.py) with
the backend name swapped.  Results are compared bit-exactly with oracle/ref_numpy.py; fp64 target of
the north star is 1e-12, fp32 target 1 ulp: both are met with equality.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BACKEND = "hip:mi300"


def _imports():
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    return gt_storage, gtscript


def lap_cartesian(inp: "Field[np.float64]", out: "Field[np.float64]"):  # noqa: F821
    with computation(PARALLEL), interval(...):  # noqa: F821
        out = -4.0 * inp[0, 0, 0] + inp[-1, 0, 0] + inp[1, 0, 0] + inp[0, -1, 0] + inp[0, 1, 0]  # noqa: F841


def avg_stencil(in_field: "Field[np.float64]", out_field: "Field[np.float64]"):  # noqa: F821
    with computation(PARALLEL), interval(...):  # noqa: F821
        out_field = 0.25 * (+in_field[0, 1, 0] + in_field[0, -1, 0] + in_field[1, 0, 0] + in_field[-1, 0, 0])  # noqa: F841


def test_storage_allocation_on_device():
    """test_interface.py:184-240 (test_allocate_gpu): alignment of the aligned_index column, layout."""
    gt_storage, _ = _imports()
    from gt4py_amd.storage import layout as gt_layout

    a = gt_storage.zeros((516, 516, 8), np.float64, backend=BACKEND, aligned_index=(2, 2, 0))
    assert isinstance(a, gt_storage.DeviceArray) and a.shape == (516, 516, 8) and a.dtype == np.float64
    # SURVEY.md Appendix E.2 with this backend's alignment -- 128 bytes, one L2 line (16 fp64 items; gt:gpu: 32): 516 -> 528 items
    assert a.strides == (8, 4224, 4224 * 516)
    rng = np.random.default_rng(0)
    for _ in range(100):
        j, k = int(rng.integers(0, 516)), int(rng.integers(0, 8))
        assert (a.ptr + 2 * 8 + j * 4224 + k * 4224 * 516) % 128 == 0
    assert gt_layout.from_name(BACKEND)["is_optimal_layout"](a, ("I", "J", "K"))
    assert a.__cuda_array_interface__["data"][0] == a.ptr and a.__hip_array_interface__["strides"] == a.strides
    f32 = gt_storage.ones((7, 5, 3), np.float32, backend=BACKEND, aligned_index=(1, 1, 0))
    assert (f32.ptr + 4) % 128 == 0 and (f32 == 1).all() and f32.sum() == 105
    host = np.arange(60.0).reshape(3, 4, 5)
    d = gt_storage.from_array(host, backend=BACKEND)
    assert np.array_equal(d.get(), host) and np.array_equal(np.asarray(d[1:, 2]), host[1:, 2])
    d[0, 0, :] = -1.0
    assert (d[0, 0] == -1).all()


def test_device_array_copy_keeps_layout_and_alignment():
    """`copy()` of a storage with padded rows: same strides, same alignment of the origin column, same values --
    so a stencil runs on the copy exactly as on the original (distributed.TunedApply calibrates on such copies)."""
    gt_storage, _ = _imports()
    from gt4py_amd.storage import layout as gt_layout

    rng = np.random.default_rng(3)
    for dtype, aligned in ((np.float32, (2, 2, 0)), (np.float64, (1, 1, 0)), (np.float64, (0, 0, 0))):
        host = rng.uniform(-1, 1, (37, 21, 5)).astype(dtype)
        a = gt_storage.from_array(host, dtype, backend=BACKEND, aligned_index=aligned)
        b = a.copy()
        assert b.strides == a.strides and b.shape == a.shape and b.dtype == a.dtype and b.ptr != a.ptr
        assert b.ptr % 512 == a.ptr % 512
        assert gt_layout.from_name(BACKEND)["is_optimal_layout"](b, ("I", "J", "K"))
        assert np.array_equal(b.get(), host)
        b[3, 4, :] = 7
        assert np.array_equal(a.get(), host)  # the copy owns its memory


def test_notebook_known_answer():
    """examples/lap_cartesian_vs_next.ipynb cells 5-9 on hip:mi300."""
    gt_storage, gtscript = _imports()
    nx = ny = 32
    lap = gtscript.stencil(backend=BACKEND, definition=lap_cartesian)
    inp = gt_storage.from_array(np.fromfunction(lambda x, y, z: x**2 + y**2, (nx, ny, 1)), backend=BACKEND,
                                aligned_index=(1, 1, 0))
    out = gt_storage.zeros((nx, ny, 1), backend=BACKEND, aligned_index=(1, 1, 0))
    lap(inp=inp, out=out, origin=(1, 1, 0), domain=(nx - 2, ny - 2, 1))
    res = out.get()
    assert (res[1:-1, 1:-1] == 4.0).all() and res.sum() == 4.0 * 30 * 30
    assert lap.backend == BACKEND and lap.field_info["inp"].boundary == ((1, 1), (1, 1), (0, 0))


def test_halo_checks_on_device():
    """test_call_interface.py:221-285 with backend hip:mi300."""
    from helpers import OriginWrapper

    gt_storage, gtscript = _imports()
    stencil = gtscript.stencil(definition=avg_stencil, backend=BACKEND)

    def pair(n=22):
        mk = lambda f: OriginWrapper(array=f(backend=BACKEND, shape=(n, n, 10), aligned_index=(1, 1, 0),  # noqa: E731
                                             dtype=np.float64), origin=(1, 1, 0))
        return mk(gt_storage.ones), mk(gt_storage.zeros)

    i, o = pair()
    stencil(in_field=i, out_field=o)
    assert (o.array[1:-1, 1:-1, :] == 1).all()
    i, o = pair()
    stencil(in_field=i, out_field=o, origin=(2, 2, 0), domain=(10, 10, 10))
    assert (o.array[2:12, 2:12, :] == 1).all() and o.array.sum() == 1000
    i, o = pair()
    with pytest.raises(ValueError):
        stencil(in_field=i, out_field=o, origin=(2, 2, 0), domain=(20, 20, 10))
    i, o = pair(23)
    stencil(in_field=i, out_field=o, origin=(2, 2, 0), domain=(20, 20, 10))


def test_host_arrays_are_rejected_and_unknown_stencils_raise():
    gt_storage, gtscript = _imports()
    stencil = gtscript.stencil(definition=avg_stencil, backend=BACKEND)
    with pytest.raises(TypeError, match="device array"):
        stencil(np.ones((8, 8, 2)), np.zeros((8, 8, 2)))

    def other(a: gtscript.Field[np.float64], b: gtscript.Field[np.float64]):
        with computation(PARALLEL), interval(...):  # noqa: F821
            t = a
            n = 0
            while n < 2:
                u = t[1, 0, 0]
                t = u + b
                n = n + 1
            a = t

    # points depend on their neighbours INSIDE one `while` body: neither the kernel library nor the
    # generic executor can run it exactly -> loud failure at decoration time, never a CPU fallback
    with pytest.raises(NotImplementedError, match="no CPU fallback"):
        gtscript.stencil(definition=other, backend=BACKEND)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_horizontal_diffusion_demo(dtype):
    """examples/cartesian/demo_horizontal_diffusion.ipynb cells 7-13 (N=30, origin (2,2,0), default domain)."""
    from gt4py_amd.cartesian.backend import hip_templates
    from oracle import ref_numpy as R

    gt_storage, gtscript = _imports()
    hd = gtscript.stencil(backend=BACKEND, definition=hip_templates.hdiff_limiter_field, dtypes={"T": dtype})
    N = 30
    idx = np.arange(N)
    xx = (np.zeros((N, N, N)) + idx.reshape(N, 1, 1)) / N
    yy = (np.zeros((N, N, N)) + idx.reshape(1, N, 1)) / N
    in_data = (5.0 + 8.0 * (2.0 + np.cos(np.pi * (xx + 1.5 * yy)) + np.sin(2 * np.pi * (xx + 1.5 * yy))) / 4.0)
    in_data = (in_data + 0.1 * np.random.default_rng(2024).uniform(-1, 1, in_data.shape)).astype(dtype)
    coeff = (0.025 * np.ones((N, N, N))).astype(dtype)
    origin = (2, 2, 0)
    d_in = gt_storage.from_array(in_data, dtype, backend=BACKEND, aligned_index=origin)
    d_out = gt_storage.from_array(np.zeros((N, N, N)), dtype, backend=BACKEND, aligned_index=origin)
    d_cf = gt_storage.from_array(coeff, dtype, backend=BACKEND, aligned_index=origin)
    exec_info = {}
    hd(d_in, d_out, d_cf, origin=origin, exec_info=exec_info)
    want = np.zeros((N, N, N), dtype)
    R.hdiff(in_data, want, coeff, domain=(N - 4, N - 4, N))
    assert np.array_equal(d_out.get(), want)
    assert exec_info["run_cpp_end_time"] >= exec_info["run_cpp_start_time"]
    assert exec_info["call_start_time"] < exec_info["run_start_time"] < exec_info["run_end_time"] < exec_info["call_end_time"]


def test_suite_horizontal_diffusion_scalar_weight():
    """TestHorizontalDiffusion (test_suites.py:200-230): definition + validation, domains 1..15, halo 2."""
    from oracle import ref_numpy as R

    gt_storage, gtscript = _imports()

    def definition(u: gtscript.Field[np.float64], diffusion: gtscript.Field[np.float64], *, weight: np.float64):
        with computation(PARALLEL), interval(...):  # noqa: F821
            laplacian = 4.0 * u[0, 0, 0] - (u[1, 0, 0] + u[-1, 0, 0] + u[0, 1, 0] + u[0, -1, 0])
            flux_i = laplacian[1, 0, 0] - laplacian[0, 0, 0]
            flux_j = laplacian[0, 1, 0] - laplacian[0, 0, 0]
            diffusion = u[0, 0, 0] - weight * (  # noqa: F841
                flux_i[0, 0, 0] - flux_i[-1, 0, 0] + flux_j[0, 0, 0] - flux_j[0, -1, 0])

    stencil = gtscript.stencil(backend=BACKEND, definition=definition)
    rng = np.random.default_rng(77)
    for domain in [(1, 1, 1), (2, 15, 3), (15, 1, 15), (15, 15, 15), (7, 9, 4)]:
        u = rng.uniform(-10, 10, (domain[0] + 4, domain[1] + 4, domain[2]))
        weight = float(rng.uniform(0, 0.5))
        d_u = gt_storage.from_array(u, backend=BACKEND, aligned_index=(2, 2, 0))
        d_out = gt_storage.zeros(domain, backend=BACKEND)
        stencil(d_u, d_out, weight=weight, origin={"u": (2, 2, 0), "diffusion": (0, 0, 0)}, domain=domain)
        got = d_out.get()
        want = R.hdiff_validation(u, weight)
        # the reference's own tolerance (suites.py:42-43) and, stronger, bit equality with the oracle
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-8)
        assert np.array_equal(got, want)


def test_tridiagonal_solver_through_the_decorator():
    from gt4py_amd.cartesian.backend import hip_templates
    from oracle import ref_numpy as R

    gt_storage, gtscript = _imports()
    tri = gtscript.stencil(backend=BACKEND, definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64})
    assert tri.domain_info.min_sequential_axis_size == 2
    rng = np.random.default_rng(7)
    shape = (33, 17, 40)
    inf, diag = rng.uniform(-1, 1, shape), rng.uniform(4, 5, shape)
    sup, rhs = rng.uniform(-1, 1, shape), rng.uniform(-10, 10, shape)
    dev = [gt_storage.from_array(a, backend=BACKEND) for a in (inf, diag, sup, rhs, np.zeros(shape))]
    tri(*dev)
    s, r, o = sup.copy(), rhs.copy(), np.zeros(shape)
    R.tridiag(inf, diag, s, r, o)
    assert np.array_equal(dev[4].get(), o) and np.array_equal(dev[2].get(), s) and np.array_equal(dev[3].get(), r)
    with pytest.raises(ValueError, match="Compute domain too small"):
        one = [gt_storage.from_array(a[:, :, :1], backend=BACKEND) for a in (inf, diag, sup, rhs, np.zeros(shape))]
        tri(*one)


def test_frozen_stencil_and_torch_tensors():
    """FrozenStencil fast path (stencil_object.py:103-136) with plain torch ROCm tensors as arguments."""
    import torch

    from oracle import ref_numpy as R

    _, gtscript = _imports()
    lap = gtscript.stencil(backend=BACKEND, definition=lap_cartesian, device_sync=False)
    rng = np.random.default_rng(1337)
    host = rng.uniform(-1, 1, (66, 34, 5))
    # I-contiguous torch tensors: allocate (K, J, I) and permute
    t_in = torch.from_numpy(np.ascontiguousarray(host.transpose(2, 1, 0))).cuda().permute(2, 1, 0)
    t_out = torch.zeros((5, 34, 66), dtype=torch.float64, device="cuda").permute(2, 1, 0)
    frozen = lap.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=(64, 32, 5))
    from gt4py_amd.storage import as_device_array

    frozen(inp=as_device_array(t_in), out=as_device_array(t_out))
    torch.cuda.synchronize()
    want = np.zeros_like(host)
    R.laplacian(host, want)
    assert np.array_equal(t_out.cpu().numpy(), want)


def test_large_domain_properties_512cubed():
    """BASELINE size (512^3 fp64): size-independent properties instead of a full CPU reference.

    * linearity of the stencil in exact arithmetic cases: lap(x^2 + y^2) == 4 everywhere,
    * lap(constant) == 0, halo of `out` untouched,
    * a checksum of a random slab against the oracle on that slab only.
    """
    from oracle import ref_numpy as R

    gt_storage, gtscript = _imports()
    import torch

    lap = gtscript.stencil(backend=BACKEND, definition=lap_cartesian)
    n = 512
    shape = (n + 2, n + 2, n)
    inp = gt_storage.empty(shape, backend=BACKEND, aligned_index=(1, 1, 0))
    out = gt_storage.full(shape, -3.0, backend=BACKEND, aligned_index=(1, 1, 0))
    x = torch.arange(n + 2, dtype=torch.float64, device="cuda").reshape(-1, 1, 1)
    y = torch.arange(n + 2, dtype=torch.float64, device="cuda").reshape(1, -1, 1)
    inp.tensor.copy_((x * x + y * y).expand(shape))
    lap(inp, out, origin=(1, 1, 0))
    t = out.tensor
    assert bool((t[1:-1, 1:-1, :] == 4.0).all())
    assert bool((t[0] == -3.0).all()) and bool((t[-1] == -3.0).all())
    assert bool((t[:, 0] == -3.0).all()) and bool((t[:, -1] == -3.0).all())
    inp.tensor.fill_(2.5)
    lap(inp, out, origin=(1, 1, 0))
    assert bool((t[1:-1, 1:-1, :] == 0.0).all())
    g = torch.Generator(device="cuda").manual_seed(1337)
    inp.tensor.copy_(torch.rand(shape, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
    lap(inp, out, origin=(1, 1, 0))
    slab = inp[:, 200:215, 100:103].get()
    want = np.zeros_like(slab)
    R.laplacian(slab, want)
    assert np.array_equal(out[1:-1, 201:214, 100:103].get(), want[1:-1, 1:-1])


def test_baseline_config1_512x512x128_with_rotating_buffer_pairs():
    """BASELINE.json configs[1] at its own size (512 x 512 x 128 fp64; one field = 285 MB with its halo, about the size of
    the 256 MB Infinity Cache), applied over FOUR rotating (inp, out) pairs the way bench.py times it (SURVEY.md section
    8d: >= 3 pairs, so that a cache-resident previous pass cannot flatter the number -- or hide a stale read here).
    Every pair: the known answer lap(x^2 + y^2) == 4 on the whole domain, the halo of `out` untouched, and a random
    field's slab against the oracle; results of earlier pairs still intact after the later launches."""
    from oracle import ref_numpy as R

    gt_storage, gtscript = _imports()
    import torch

    lap = gtscript.stencil(backend=BACKEND, definition=lap_cartesian, device_sync=False)
    dom = (512, 512, 128)
    shape = (dom[0] + 2, dom[1] + 2, dom[2])
    origin = {"inp": (1, 1, 0), "out": (1, 1, 0)}
    frozen = lap.freeze(origin=origin, domain=dom)
    x = torch.arange(shape[0], dtype=torch.float64, device="cuda").reshape(-1, 1, 1)
    y = torch.arange(shape[1], dtype=torch.float64, device="cuda").reshape(1, -1, 1)
    g = torch.Generator(device="cuda").manual_seed(1337)
    pairs = []
    for n in range(4):
        inp = gt_storage.empty(shape, backend=BACKEND, aligned_index=(1, 1, 0))
        out = gt_storage.full(shape, -3.0, backend=BACKEND, aligned_index=(1, 1, 0))
        if n % 2 == 0:
            inp.tensor.copy_(((x * x + y * y) * float(n + 1)).expand(shape))
        else:
            inp.tensor.copy_(torch.rand(shape, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
        pairs.append((inp, out))
    for rounds in range(3):  # the rotation of the benchmark: pair 0, 1, 2, 3, 0, 1, ...
        for inp, out in pairs:
            frozen(inp=inp, out=out)
    torch.cuda.synchronize()
    for n, (inp, out) in enumerate(pairs):
        t = out.tensor
        assert bool((t[0] == -3.0).all()) and bool((t[-1] == -3.0).all()) and bool((t[:, 0] == -3.0).all()) and bool((t[:, -1] == -3.0).all())
        if n % 2 == 0:
            assert bool((t[1:-1, 1:-1, :] == 4.0 * (n + 1)).all())
        else:
            slab = inp[:, 300:318, 60:64].get()
            want = np.zeros_like(slab)
            R.laplacian(slab, want)
            assert np.array_equal(out[1:-1, 301:317, 60:64].get(), want[1:-1, 1:-1])


def test_large_domain_properties_hdiff_and_tridiagonal():
    """BASELINE sizes C3 (1024x1024x80 fp32 hdiff) and C4 (1024x1024x160 fp64 tridiagonal): properties
    that need no full-size CPU run.

    * hdiff of an affine plane is the identity (the Laplacian of a plane is exactly 0 in these units);
    * hdiff of a random field equals the oracle on a slab cut out with its halo;
    * the tridiagonal solution satisfies the ORIGINAL system: residual computed on the device with torch,
      and two columns are compared bit for bit with the oracle.
    """
    from oracle import ref_numpy as R

    gt_storage, gtscript = _imports()
    import torch
    from gt4py_amd.cartesian.backend import hip_templates

    hd = gtscript.stencil(backend=BACKEND, definition=hip_templates.hdiff_limiter_field, dtypes={"T": np.float32})
    ni, nj, nk = 1024, 1024, 80
    shape = (ni + 4, nj + 4, nk)
    mk = lambda: gt_storage.empty(shape, np.float32, backend=BACKEND, aligned_index=(2, 2, 0))  # noqa: E731
    inp, coeff, out = mk(), mk(), mk()
    x = torch.arange(shape[0], dtype=torch.float32, device="cuda").reshape(-1, 1, 1)
    y = torch.arange(shape[1], dtype=torch.float32, device="cuda").reshape(1, -1, 1)
    inp.tensor.copy_((3.0 * x - 2.0 * y + 7.0).expand(shape))  # small integers: exact in fp32
    coeff.tensor.fill_(0.025)
    out.tensor.fill_(-1.0)
    hd(inp, out, coeff, origin=(2, 2, 0))
    assert bool((out.tensor[2:-2, 2:-2] == inp.tensor[2:-2, 2:-2]).all())
    assert bool((out.tensor[:2] == -1.0).all()) and bool((out.tensor[:, -2:] == -1.0).all())
    g = torch.Generator(device="cuda").manual_seed(2024)
    inp.tensor.copy_(torch.rand(shape, dtype=torch.float32, device="cuda", generator=g) * 10)
    hd(inp, out, coeff, origin=(2, 2, 0))
    sl = (slice(500, 540), slice(1000, 1028), slice(30, 33))
    h_in, h_co = inp[sl].get(), coeff[sl].get()
    want = np.zeros_like(h_in)
    R.hdiff(h_in, want, h_co, origin_in=(2, 2, 0), origin_out=(2, 2, 0), origin_coeff=(2, 2, 0),
            domain=(h_in.shape[0] - 4, h_in.shape[1] - 4, 3))
    assert np.array_equal(out[sl].get()[2:-2, 2:-2], want[2:-2, 2:-2])
    del inp, coeff, out

    tri = gtscript.stencil(backend=BACKEND, definition=hip_templates.tridiagonal_solver, dtypes={"T": np.float64})
    dom = (1024, 1024, 160)
    f = {}
    for name, (lo, hi) in {"inf": (-1, 1), "diag": (4, 5), "sup": (-1, 1), "rhs": (-10, 10), "out": (0, 0)}.items():
        f[name] = gt_storage.empty(dom, np.float64, backend=BACKEND)
        f[name].tensor.copy_(torch.rand(dom, dtype=torch.float64, device="cuda", generator=g) * (hi - lo) + lo)
    sup0, rhs0 = f["sup"].tensor.clone(), f["rhs"].tensor.clone()
    cols = [(0, 0), (1023, 517)]
    host = {n: np.stack([f[n][i, j, :].get() for i, j in cols])[:, None, :].copy() for n in f}
    tri(**f)
    xs = f["out"].tensor
    res = f["diag"].tensor * xs - rhs0
    res[:, :, 1:] += f["inf"].tensor[:, :, 1:] * xs[:, :, :-1]
    res[:, :, :-1] += sup0[:, :, :-1] * xs[:, :, 1:]
    assert float(res.abs().max()) < 1e-13 * 10 * 8
    R.tridiag(host["inf"], host["diag"], host["sup"], host["rhs"], host["out"])
    for n, (i, j) in enumerate(cols):
        for name in ("sup", "rhs", "out"):
            assert np.array_equal(f[name][i, j, :].get(), host[name][n, 0]), (name, i, j)


def test_stencil_calls_can_be_captured_in_a_hip_graph():
    """Launch-bound loops belong in hipGraphs: both execution paths launch on the caller's current stream, so
    `torch.cuda.graph` (hipGraph on ROCm) captures them; replays reproduce the eager result bit for bit."""
    import torch
    from oracle import ref_numpy as R

    gt_storage, gtscript = _imports()

    def smooth(a: gtscript.Field[np.float64], b: gtscript.Field[np.float64], *, w: float):
        with computation(PARALLEL), interval(...):  # noqa: F821
            t = a[1, 0, 0] + a[-1, 0, 0] + a[0, 1, 0] + a[0, -1, 0]
            b = a + w * (t - 4.0 * a)  # noqa: F841

    lap = gtscript.stencil(backend=BACKEND, definition=lap_cartesian, device_sync=False)  # kernel library
    gen = gtscript.stencil(backend=BACKEND, definition=smooth, device_sync=False)  # generated kernel
    rng = np.random.default_rng(3)
    host = rng.uniform(-1, 1, (66, 34, 6))
    a = gt_storage.from_array(host, backend=BACKEND, aligned_index=(1, 1, 0))
    b = gt_storage.zeros(host.shape, backend=BACKEND, aligned_index=(1, 1, 0))
    c = gt_storage.zeros(host.shape, backend=BACKEND, aligned_index=(1, 1, 0))
    dom = (64, 32, 6)
    lap_f = lap.freeze(origin={"inp": (1, 1, 0), "out": (1, 1, 0)}, domain=dom)
    gen_f = gen.freeze(origin={"a": (1, 1, 0), "b": (1, 1, 0)}, domain=dom)

    def sequence():
        lap_f(inp=a, out=b)
        gen_f(a=b, b=c, w=0.125)

    sequence()  # eager: compiles, fills caches, gives the expected values
    torch.cuda.synchronize()
    want_b, want_c = b.get().copy(), c.get().copy()
    chk = np.zeros_like(host)
    R.laplacian(host, chk)
    assert np.array_equal(want_b, chk)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sequence()
    for fill in (7.0, -3.0):
        b.tensor.fill_(fill)
        c.tensor.fill_(fill)
        graph.replay()
        torch.cuda.synchronize()
        got_b, got_c = b.get(), c.get()
        assert np.array_equal(got_b[1:-1, 1:-1], want_b[1:-1, 1:-1])
        assert np.array_equal(got_c[2:-2, 2:-2], want_c[2:-2, 2:-2])


def column_sum(a: "Field[np.float64]", s: "Field[np.float64]"):  # noqa: F821
    with computation(FORWARD):  # noqa: F821
        with interval(0, 1):  # noqa: F821
            s = a  # noqa: F841
        with interval(1, None):  # noqa: F821
            s = s[0, 0, -1] + a  # noqa: F841


def test_an_empty_axis_is_a_call_that_writes_nothing():
    """The reference refuses only the ALL-zero domain (stencil_object.py:370-373 with the partial order of
    gtc/definitions.py:141-171); a domain with one empty axis passes validation and the numpy backend's slices are
    empty.  Same here, through the hand-written kernels, a generated horizontal kernel and a generated column
    kernel: no launch, no write, no error."""
    gt_storage, gtscript = _imports()
    lap = gtscript.stencil(backend=BACKEND, definition=lap_cartesian)        # kernel library
    avg = gtscript.stencil(backend=BACKEND, definition=avg_stencil)          # generated, horizontal
    col = gtscript.stencil(backend=BACKEND, definition=column_sum)           # generated, column
    mk = lambda fill: gt_storage.full((12, 12, 6), fill, backend=BACKEND, aligned_index=(1, 1, 0), dtype=np.float64)  # noqa: E731
    for stencil in (lap, avg, col):
        for domain in ((0, 4, 4), (4, 0, 4), (4, 4, 0)):
            a, b = mk(1.0), mk(7.0)
            if stencil is col and domain[2] == 0:  # its two intervals need two levels (min_sequential_axis_size)
                with pytest.raises(ValueError, match="Compute domain too small. Sequential axis is 0"):
                    stencil(a, b, origin=(1, 1, 0), domain=domain)
            else:
                stencil(a, b, origin=(1, 1, 0), domain=domain)
            assert (b.get() == 7.0).all() and (a.get() == 1.0).all(), (stencil.options["name"], domain)
        with pytest.raises(ValueError, match="Compute domain contains zero sizes"):
            stencil(mk(1.0), mk(7.0), origin=(1, 1, 0), domain=(0, 0, 0))


@pytest.mark.multiprocess
def test_the_headline_line_of_bench_py_carries_what_the_contract_and_the_verdicts_ask_for():
    """`python bench.py` at N = 1 (short: 3 steps, no CPU baseline, no other kernels): ONE JSON line on stdout with the contract's keys,
    the `roofline` object, the figures a drop-in user gets (`value_default_allocator`, `value_allocator_off`), and -- as the LAST key, so
    that a record which keeps only the tail of a long line keeps it -- the `summary` of every config's figure."""
    import json
    import os
    import pathlib
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parent.parent
    proc = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-kernels"],
                          capture_output=True, text=True, timeout=600, cwd=str(root), env=dict(os.environ))
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1  # the contract: one line
    line = json.loads(lines[0])
    for key, want in (("n_gpus", 1), ("steps", 3), ("warmup", 1), ("higher_is_better", True), ("dtype", "f64"), ("unit", "GLUPS")):
        assert line[key] == want, key
    assert line["scaling"] == "strong"  # the 512^3 grid is FIXED and split over the ranks (hdiff2048 is the weak-scaling workload)
    assert "512" in line["metric"] and "workload" in line["config"] and line["vs_baseline"] is None
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert 0.5 < roof["frac"] < 1.0 and abs(line["value"] * 16.0 - roof["achieved"]) / roof["achieved"] < 0.1  # GLUPS x 16 B ~ GB/s
    assert line["value_default_allocator"] > 0 and line["value_allocator_off"] > 0 and "allocator_note" in line
    assert list(line)[-1] == "summary" and line["summary"]["lap5_f64_512"]["glups"] == line["value"]
    assert line["memory_groups"]["enabled"] in (True, False) and "fields" not in line["memory_groups"]
