"""Feature tests of the reference around the call path: ``exec_info`` bookkeeping, foreign array layouts, lazy stencils.

``exec_info`` bookkeeping of stencil calls, as /root/reference/tests/cartesian_tests/integration_tests/
feature_tests/test_exec_info.py pins it: per-call timestamps, the normalised origin / domain, and the per-stencil
aggregates kept when ``exec_info["__aggregate_data"]`` is set.  Same two stencils (upwind advection, fourth-order
diffusion), same call sequence; sizes are fixed instead of drawn by hypothesis.  Oracle on the CPU, ``hip:mi300`` on the
GPU (where the native launch timestamps ``run_cpp_*`` must be there as well)."""

import numpy as np
import pytest

import gt4py_amd.storage as gt_storage
import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
from gt4py_amd.cartesian import gtscript
from gt4py_amd.cartesian.gtscript import PARALLEL, Field, computation, interval  # noqa: F401

BACKENDS = ["numpy", pytest.param("hip:mi300", marks=pytest.mark.gpu)]


def advection_def(in_phi: Field[float], in_u: Field[float], in_v: Field[float], out_phi: Field[float]):
    with computation(PARALLEL), interval(...):
        u = 0.5 * (in_u[-1, 0, 0] + in_u[0, 0, 0])
        flux_x = u[0, 0, 0] * (in_phi[-1, 0, 0] if u[0, 0, 0] > 0 else in_phi[0, 0, 0])
        v = 0.5 * (in_v[0, -1, 0] + in_v[0, 0, 0])
        flux_y = v[0, 0, 0] * (in_phi[0, -1, 0] if v[0, 0, 0] > 0 else in_phi[0, 0, 0])
        out_phi = in_phi - (flux_x[1, 0, 0] - flux_x[0, 0, 0]) - (flux_y[0, 1, 0] - flux_y[0, 0, 0])


def diffusion_def(in_phi: Field[float], out_phi: Field[float], *, alpha: float):
    with computation(PARALLEL), interval(...):
        lap1 = -4 * in_phi[0, 0, 0] + in_phi[-1, 0, 0] + in_phi[1, 0, 0] + in_phi[0, -1, 0] + in_phi[0, 1, 0]
        lap2 = -4 * lap1[0, 0, 0] + lap1[-1, 0, 0] + lap1[1, 0, 0] + lap1[0, -1, 0] + lap1[0, 1, 0]
        flux_x = lap2[1, 0, 0] - lap2[0, 0, 0]
        flux_y = lap2[0, 1, 0] - lap2[0, 0, 0]
        out_phi = in_phi + alpha * (flux_x[0, 0, 0] - flux_x[-1, 0, 0] + flux_y[0, 0, 0] - flux_y[0, -1, 0])


NX, NY, NZ = 17, 11, 6


def _setup(backend):
    rng = np.random.default_rng(5)
    advection = gtscript.stencil(backend=backend, definition=advection_def)
    diffusion = gtscript.stencil(backend=backend, definition=diffusion_def)

    def field(aligned):
        return gt_storage.from_array(rng.uniform(-1, 1, (NX, NY, NZ)), backend=backend, aligned_index=aligned, dtype=float)

    fields = {"in_phi": field((0, 0, 0)), "in_u": field((0, 0, 0)), "in_v": field((0, 0, 0)), "tmp_phi": field((1, 1, 0)),
              "out_phi": field((3, 3, 0))}
    return advection, diffusion, fields


def _run(advection, diffusion, f, exec_info, nt):
    for _ in range(nt):
        advection(f["in_phi"], f["in_u"], f["in_v"], f["tmp_phi"], origin=(1, 1, 0), domain=(NX - 2, NY - 2, NZ),
                  exec_info=exec_info)
        diffusion(f["in_phi"], f["out_phi"], alpha=1 / 32, origin=(3, 3, 0), domain=(NX - 6, NY - 6, NZ), exec_info=exec_info)


def _check_exec_info(exec_info, native):
    """test_exec_info.py:119-145: the entries of the LAST call."""
    assert exec_info["call_end_time"] > exec_info["call_start_time"]
    assert exec_info["run_start_time"] > exec_info["call_start_time"]
    assert exec_info["run_end_time"] > exec_info["run_start_time"]
    assert exec_info["call_end_time"] > exec_info["run_end_time"]
    if native:
        assert exec_info["run_cpp_end_time"] >= exec_info["run_cpp_start_time"]
        # device-side interval of the call's kernels (hipEvent pair), nested in the run bracket like run_cpp_*
        assert exec_info["run_start_time"] <= exec_info["run_cpp_start_time"] <= exec_info["run_hip_start_time"]
        assert exec_info["run_hip_start_time"] < exec_info["run_hip_end_time"] <= exec_info["run_end_time"]
        assert 1e-6 < exec_info["run_hip_end_time"] - exec_info["run_hip_start_time"] < 1.0
    assert exec_info["origin"] == {"_all_": (3, 3, 0), "in_phi": (3, 3, 0), "out_phi": (3, 3, 0)}
    assert exec_info["domain"] == (NX - 6, NY - 6, NZ)


def _check_stencil_info(exec_info, info, nt, native, last_called_stencil=False):
    """test_exec_info.py:147-192: the aggregates of one stencil."""
    assert info["ncalls"] == nt
    assert info["call_end_time"] > info["call_start_time"]
    assert np.isclose(info["call_time"], info["call_end_time"] - info["call_start_time"])
    assert info["total_call_time"] == info["call_time"] if nt == 1 else info["total_call_time"] > info["call_time"]
    if last_called_stencil:
        assert info["call_start_time"] == exec_info["call_start_time"] and info["call_end_time"] == exec_info["call_end_time"]
        assert np.isclose(info["run_time"], exec_info["run_end_time"] - exec_info["run_start_time"])
    assert info["call_time"] > info["run_time"]
    assert info["total_run_time"] == info["run_time"] if nt == 1 else info["total_run_time"] > info["run_time"]
    if native:
        if last_called_stencil:
            assert np.isclose(info["run_cpp_time"], exec_info["run_cpp_end_time"] - exec_info["run_cpp_start_time"])
        assert info["run_time"] > info["run_cpp_time"]
        assert info["total_run_cpp_time"] == info["run_cpp_time"] if nt == 1 else info["total_run_cpp_time"] > info["run_cpp_time"]
        if last_called_stencil:
            assert np.isclose(info["run_hip_time"], exec_info["run_hip_end_time"] - exec_info["run_hip_start_time"])
        assert info["run_time"] > info["run_hip_time"] > 0.0
        assert info["total_run_hip_time"] == info["run_hip_time"] if nt == 1 else info["total_run_hip_time"] > info["run_hip_time"]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("nt", [1, 3])
def test_backcompatibility(backend, nt):
    advection, diffusion, fields = _setup(backend)
    exec_info = {}
    _run(advection, diffusion, fields, exec_info, nt)
    _check_exec_info(exec_info, native=backend != "numpy")
    assert exec_info["__aggregate_data"] is False
    assert type(advection).__name__ not in exec_info and type(diffusion).__name__ not in exec_info


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("nt", [1, 3])
def test_aggregate(backend, nt):
    advection, diffusion, fields = _setup(backend)
    exec_info = {"__aggregate_data": True}
    _run(advection, diffusion, fields, exec_info, nt)
    native = backend != "numpy"
    _check_exec_info(exec_info, native)
    _check_stencil_info(exec_info, exec_info[type(advection).__name__], nt, native)
    _check_stencil_info(exec_info, exec_info[type(diffusion).__name__], nt, native, last_called_stencil=True)


@pytest.mark.parametrize("backend", BACKENDS)
def test_results_of_the_two_stencils(backend):
    """... and the values: oracle == hip:mi300 is covered by the suites; here the two backends see the same call."""
    advection, diffusion, fields = _setup(backend)
    _run(advection, diffusion, fields, None, 1)
    ref_adv, ref_diff, ref_fields = _setup("numpy")
    _run(ref_adv, ref_diff, ref_fields, None, 1)
    for name in ("tmp_phi", "out_phi"):
        np.testing.assert_array_equal(gt_storage.asnumpy(fields[name]), np.asarray(ref_fields[name]))


# ---- test_field_layouts.py:33-95 (arrays that did not come from gt4py.storage) -----------------------------------
def copy_stencil(field_a: Field[float], field_b: Field[float]):
    with computation(PARALLEL), interval(...):
        field_b = field_a


@pytest.mark.parametrize("order", ["C", "F"])
def test_numpy_allocators(order):
    rng = np.random.default_rng(0)
    inp = np.array(rng.standard_normal((20, 10, 5)), order=order, dtype=np.float64)
    outp = np.zeros((20, 10, 5), order=order, dtype=np.float64)
    gtscript.stencil(definition=copy_stencil, backend="numpy")(field_a=inp, field_b=outp)
    np.testing.assert_array_equal(outp, inp)


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["C", "F"])
def test_foreign_device_arrays_of_either_order(order):
    """The reference skips this on ROCm (no `__hip_array_interface__` on cupy arrays there); here torch tensors of any
    stride order are accepted as they are (any-stride kernels), with the layout warning for the non-preferred one."""
    import warnings

    import torch

    host = np.random.default_rng(0).standard_normal((20, 10, 5))
    inp = torch.as_tensor(host, device="cuda")
    outp = torch.zeros((20, 10, 5), dtype=torch.float64, device="cuda")
    if order == "F":  # I contiguous: the layout hip:mi300 prefers
        inp = inp.permute(2, 1, 0).contiguous().permute(2, 1, 0)
        outp = outp.permute(2, 1, 0).contiguous().permute(2, 1, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        gtscript.stencil(definition=copy_stencil, backend="hip:mi300")(field_a=inp, field_b=outp)
    np.testing.assert_array_equal(outp.cpu().numpy(), host)


def test_bad_layout_warns():
    inp = np.transpose(np.random.default_rng(1).standard_normal((10, 10, 10)), axes=(1, 2, 0))
    outp = gt_storage.zeros(backend="numpy", shape=(10, 10, 10), dtype=np.float64, aligned_index=(0, 0, 0))
    with pytest.warns(UserWarning, match="The layout of the field 'field_a' is not recommended for this backend."
                                         "This may lead to performance degradation. Please consider using the"
                                         "provided allocators in `gt4py.storage`."):
        gtscript.stencil(definition=copy_stencil, backend="numpy")(field_a=inp, field_b=outp)


@pytest.mark.parametrize("backend", ["numpy", "gt:cpu_ifirst", "gt:cpu_kfirst", pytest.param("hip:mi300", marks=pytest.mark.gpu)])
def test_data_dimensions_stride_is_always_higher_than_cartesian(backend):
    a4 = gt_storage.zeros(backend=backend, shape=(2, 2, 2, 2), dtype=np.float64, aligned_index=(0, 0, 0, 0))
    assert a4.strides[3] > max(a4.strides[0:3])
    a5 = gt_storage.zeros(backend=backend, shape=(2, 2, 2, 2, 2), dtype=np.float64, aligned_index=(0, 0, 0, 0, 0))
    assert a5.strides[4] > max(a5.strides[0:3]) and a5.strides[3] > max(a5.strides[0:3])


# ---- test_gtcnumpy.py:14-33 and unit_tests/test_lazy_stencil.py:52-90 ---------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_masked_vector_assignment(backend):
    from gt4py_amd.cartesian.gtscript import FORWARD, IJ

    @gtscript.stencil(backend)
    def masked_vector_assignment(fld2D: Field[IJ, np.float64]):
        with computation(FORWARD), interval(0, None):
            fld2D += fld2D
            if fld2D >= 1.0:
                fld2D = 0.0

    fld2D = gt_storage.ones(shape=(2, 3), dtype=np.float64, backend=backend, aligned_index=(0, 0), dimensions="IJ")
    masked_vector_assignment(fld2D, domain=(2, 3, 4))
    assert (gt_storage.asnumpy(fld2D) == 0).all()


def copy_stencil_definition(out_f: Field[float], in_f: Field[float]):
    with computation(PARALLEL), interval(...):
        out_f = in_f


def wrong_syntax_stencil_definition(out_f: Field[float], in_f: Field[float]):
    from __externals__ import undefined

    with computation(PARALLEL), interval(...):
        out_f = undefined(in_f)


@pytest.mark.parametrize("backend", BACKENDS)
def test_lazy_stencil_builds_on_first_use_and_is_callable(backend):
    from gt4py_amd.cartesian.definitions import GTScriptDefinitionError

    lazy = gtscript.lazy_stencil(backend=backend, definition=copy_stencil_definition, rebuild=True)
    assert lazy.backend == backend  # building it now
    a = gt_storage.from_array(np.array([[[1.0]]]), aligned_index=(0, 0, 0), backend=backend, dtype=float)
    b = gt_storage.from_array(np.array([[[0.0]]]), aligned_index=(0, 0, 0), backend=backend, dtype=float)
    lazy(b, a)
    assert gt_storage.asnumpy(b)[0, 0, 0] == 1.0
    # a definition with a GTScript error: nothing happens until it is needed, then the frontend's error
    broken = gtscript.lazy_stencil(backend=backend, definition=wrong_syntax_stencil_definition)
    with pytest.raises(GTScriptDefinitionError):
        broken.implementation  # noqa: B018


@pytest.mark.gpu
def test_exec_info_device_time_scales_with_the_work():
    """``run_hip_*`` is time on the DEVICE: for the 512^3 Laplacian it is the kernel's 0.3-0.5 ms, while ``run_cpp_*``
    (an asynchronous launch) stays in the tens of microseconds; without ``exec_info`` no event is recorded and the call
    does not wait (covered by the hipGraph capture test, which would fail on a synchronising call)."""
    from gt4py_amd.cartesian.backend import hip_templates

    lap = gtscript.stencil(backend="hip:mi300", definition=hip_templates.lap_notebook, dtypes={"T": np.float64}, device_sync=False)
    big = gt_storage.zeros((514, 514, 512), backend="hip:mi300", aligned_index=(1, 1, 0))
    out = gt_storage.zeros((514, 514, 512), backend="hip:mi300", aligned_index=(1, 1, 0))
    small = gt_storage.zeros((34, 34, 4), backend="hip:mi300", aligned_index=(1, 1, 0))
    out_small = gt_storage.zeros((34, 34, 4), backend="hip:mi300", aligned_index=(1, 1, 0))
    times = {}
    for name, (a, b) in (("big", (big, out)), ("small", (small, out_small))):
        best = None
        for _ in range(5):
            info = {}
            lap(a, b, origin=(1, 1, 0), exec_info=info)
            hip = info["run_hip_end_time"] - info["run_hip_start_time"]
            cpp = info["run_cpp_end_time"] - info["run_cpp_start_time"]
            best = (hip, cpp) if best is None or hip < best[0] else best
        times[name] = best
    assert 2.5e-4 < times["big"][0] < 2e-3, times           # ~0.36 ms of kernel
    assert times["small"][0] < times["big"][0] / 5, times    # a 32 x 32 x 4 domain is launch latency only
    assert times["big"][1] < times["big"][0], times          # the launch call returns before the kernel has finished
