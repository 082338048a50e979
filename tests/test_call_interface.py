"""Stencil call interface (host logic) on the CPU, executed through the ORACLE numpy backend.

The argument handling of ``StencilObject`` (origins, domains, validation, caches, exec_info,
freeze) is backend independent; the reference pins it in
/root/reference/tests/cartesian_tests/integration_tests/feature_tests/test_call_interface.py,
test_stencil_object.py, test_field_layouts.py and test_exec_info.py.  The same scenarios and exact
expected values are checked here.  ``backend="numpy"`` below is oracle/numpy_backend.py (test
infrastructure); the GPU counterparts live in tests/test_gpu_stencils.py.
"""

import copy
import warnings

import numpy as np
import pytest

import oracle.numpy_backend  # noqa: F401  registers backend "numpy" (oracle, tests only)
import gt4py_amd.storage as gt_storage
from gt4py_amd.cartesian import backend as gt_backend, gtscript
from gt4py_amd.cartesian.gtscript import FORWARD, PARALLEL, Field, K, computation, interval  # noqa: F401
from helpers import DimensionsWrapper, OriginWrapper


def base_stencil(field1: Field[np.float64], field2: Field[np.float64], field3: Field[np.float32], *, param: np.float64):
    with computation(PARALLEL), interval(...):
        field1 = field2 + field3 * param
        field2 = field1 + field3 * param
        field3 = param * field2


def _abc():
    A = gt_storage.ones(backend="gt:cpu_ifirst", dtype=np.float64, shape=(3, 3, 3), aligned_index=(0, 0, 0))
    B = gt_storage.ones(backend="gt:cpu_kfirst", dtype=np.float64, shape=(3, 3, 3), aligned_index=(2, 2, 2))
    C = gt_storage.ones(backend="numpy", dtype=np.float32, shape=(3, 3, 3), aligned_index=(0, 1, 0))
    return A, B, C


def test_origin_selection():
    """test_call_interface.py:36-103: explicit origin > '_all_' > per-array __gt_origin__."""
    stencil = gtscript.stencil(definition=base_stencil, backend="numpy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # mixed layouts on purpose

        A, B, C = _abc()
        wraps = (OriginWrapper(array=A, origin=(0, 0, 0)), OriginWrapper(array=B, origin=(2, 2, 2)),
                 OriginWrapper(array=C, origin=(0, 1, 0)))
        stencil(*wraps, param=3.0, origin=(1, 1, 1), domain=(1, 1, 1))
        assert A[1, 1, 1] == 4 and B[1, 1, 1] == 7 and C[1, 1, 1] == 21
        assert np.sum(A) == 30 and np.sum(B) == 33 and np.sum(C) == 47

        A, B, C = _abc()
        wraps = (OriginWrapper(array=A, origin=(0, 0, 0)), OriginWrapper(array=B, origin=(2, 2, 2)),
                 OriginWrapper(array=C, origin=(0, 1, 0)))
        stencil(*wraps, param=3.0, origin={"_all_": (1, 1, 1), "field1": (2, 2, 2)}, domain=(1, 1, 1))
        assert A[2, 2, 2] == 4 and B[1, 1, 1] == 7 and C[1, 1, 1] == 21
        assert np.sum(A) == 30 and np.sum(B) == 33 and np.sum(C) == 47

        A, B, C = _abc()
        wraps = (OriginWrapper(array=A, origin=(0, 0, 0)), OriginWrapper(array=B, origin=(2, 2, 2)),
                 OriginWrapper(array=C, origin=(0, 1, 0)))
        stencil(*wraps, param=3.0, origin={"field1": (2, 2, 2)}, domain=(1, 1, 1))
        assert A[2, 2, 2] == 4 and B[2, 2, 2] == 7 and C[0, 1, 0] == 21
        assert np.sum(A) == 30 and np.sum(B) == 33 and np.sum(C) == 47


def test_domain_selection():
    """test_call_interface.py:106-137."""
    stencil = gtscript.stencil(definition=base_stencil, backend="numpy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        A, B, C = _abc()
        stencil(A, B, C, param=3.0, origin=(1, 1, 1), domain=(1, 1, 1))
        assert A[1, 1, 1] == 4 and B[1, 1, 1] == 7 and C[1, 1, 1] == 21
        assert np.sum(A) == 30 and np.sum(B) == 33 and np.sum(C) == 47
        A, B, C = _abc()
        stencil(A, B, C, param=3.0, origin=(0, 0, 0))
        assert np.all(A == 4) and np.all(B == 7) and np.all(C == 21)


def a_stencil(arg1: Field[np.float64], arg2: Field[np.float64], arg3: Field[np.float64] = None, *,
              par1: np.float64, par2: np.float64 = 7.0, par3: np.float64 = None):
    from __externals__ import BRANCH

    with computation(PARALLEL), interval(...):
        if __INLINED(BRANCH):  # noqa: F821
            arg1 = arg1 * par1 * par2
        else:
            arg1 = arg2 + arg3 * par1 * par2 * par3


def test_default_arguments():
    """test_call_interface.py:167-218."""
    backend = "numpy"
    branch_true = gtscript.stencil(backend=backend, definition=a_stencil, externals={"BRANCH": True}, rebuild=True)
    branch_false = gtscript.stencil(backend=backend, definition=a_stencil, externals={"BRANCH": False}, rebuild=True)
    mk = lambda f: f(backend=backend, dtype=np.float64, shape=(3, 3, 3), aligned_index=(0, 0, 0))  # noqa: E731
    arg1, arg2, arg3 = mk(gt_storage.ones), mk(gt_storage.zeros), mk(gt_storage.ones)
    arg3 *= 2

    branch_true(arg1, None, arg3, par1=2.0)
    np.testing.assert_equal(arg1, 14 * np.ones((3, 3, 3)))
    branch_true(arg1, None, par1=2.0)
    np.testing.assert_equal(arg1, 196 * np.ones((3, 3, 3)))
    branch_false(arg1, arg2, arg3, par1=2.0, par3=2.0)
    np.testing.assert_equal(arg1, 56 * np.ones((3, 3, 3)))
    with pytest.raises((ValueError, AssertionError)):
        branch_false(arg1, arg2, par1=2.0, par3=2.0)

    arg1, arg2, arg3 = mk(gt_storage.ones), mk(gt_storage.zeros), mk(gt_storage.ones)
    arg3 *= 2
    branch_true(arg1, arg2=None, par1=2.0, par2=5.0, par3=3.0)
    np.testing.assert_equal(arg1, 10 * np.ones((3, 3, 3)))
    branch_true(arg1, arg2=None, par1=2.0, par2=5.0)
    np.testing.assert_equal(arg1, 100 * np.ones((3, 3, 3)))
    branch_false(arg1, arg2, arg3, par1=2.0, par2=5.0, par3=3.0)
    np.testing.assert_equal(arg1, 60 * np.ones((3, 3, 3)))
    with pytest.raises((TypeError, AssertionError)):
        branch_false(arg1, arg2, arg3, par1=2.0, par2=5.0)


def avg_stencil(in_field: Field[np.float64], out_field: Field[np.float64]):
    with computation(PARALLEL), interval(...):
        out_field = 0.25 * (+in_field[0, 1, 0] + in_field[0, -1, 0] + in_field[1, 0, 0] + in_field[-1, 0, 0])


def _in_out(backend, n=22):
    mk = lambda f: OriginWrapper(  # noqa: E731
        array=f(backend=backend, shape=(n, n, 10), aligned_index=(1, 1, 0), dtype=np.float64), origin=(1, 1, 0))
    return mk(gt_storage.ones), mk(gt_storage.zeros)


@pytest.mark.parametrize("backend", ["numpy"])
def test_halo_checks(backend):
    """test_call_interface.py:221-285."""
    stencil = gtscript.stencil(definition=avg_stencil, backend=backend)
    in_field, out_field = _in_out(backend)
    stencil(in_field=in_field, out_field=out_field)
    assert (out_field.array[1:-1, 1:-1, :] == 1).all()

    in_field, out_field = _in_out(backend)
    stencil(in_field=in_field, out_field=out_field, origin=(2, 2, 0), domain=(10, 10, 10))
    assert (out_field.array[2:12, 2:12, :] == 1).all()
    assert out_field.array.sum() == 1000

    in_field, out_field = _in_out(backend)
    with pytest.raises(ValueError):
        stencil(in_field=in_field, out_field=out_field, origin=(2, 2, 0), domain=(20, 20, 10))

    in_field, out_field = _in_out(backend, n=23)
    stencil(in_field=in_field, out_field=out_field, origin=(2, 2, 0), domain=(20, 20, 10))


def test_np_int_types():
    """test_call_interface.py:288-311."""
    stencil = gtscript.stencil(definition=avg_stencil, backend="numpy")
    in_field = gt_storage.ones(backend="numpy", shape=(np.int8(23), np.int16(23), np.int32(10)),
                               aligned_index=(np.int64(1), int(1), 0), dtype=np.float64)
    out_field = gt_storage.zeros(backend="numpy", shape=(np.int8(23), np.int16(23), np.int32(10)),
                                 aligned_index=(np.int64(1), int(1), 0), dtype=np.float64)
    stencil(in_field=in_field, out_field=out_field, origin=(np.int8(2), np.int16(2), np.int32(0)),
            domain=(np.int64(20), int(20), 10))
    assert (out_field[2:22, 2:22] == 1).all()


def test_exec_info():
    """test_call_interface.py:314-343 and test_exec_info.py:119-134 (key set and ordering)."""
    stencil = gtscript.stencil(definition=avg_stencil, backend="numpy")
    exec_info = {}
    in_field = gt_storage.ones(backend="numpy", shape=(23, 23, 10), aligned_index=(1, 1, 0), dtype=np.float64)
    out_field = gt_storage.zeros(backend="numpy", shape=(23, 23, 10), aligned_index=(1, 1, 0), dtype=np.float64)
    stencil(in_field=in_field, out_field=out_field, origin=(2, 2, 0), domain=(20, 20, 10), exec_info=exec_info)
    for k in ("call", "call_run", "run"):
        assert exec_info[k + "_end_time"] > exec_info[k + "_start_time"]
    assert exec_info["call_start_time"] < exec_info["call_run_start_time"] < exec_info["run_start_time"]
    assert exec_info["run_end_time"] < exec_info["call_run_end_time"] < exec_info["call_end_time"]
    # aggregated counters (stencil_module.py.in:125-158)
    agg = {"__aggregate_data": True}
    for _ in range(3):
        stencil(in_field=in_field, out_field=out_field, origin=(2, 2, 0), domain=(20, 20, 10), exec_info=agg)
    stats = agg[type(stencil).__name__]
    assert stats["ncalls"] == 3 and stats["total_call_time"] >= stats["call_time"] > 0
    assert stats["total_run_time"] >= stats["run_time"] > 0


class TestAxesMismatch:
    """test_call_interface.py:346-380."""

    @pytest.fixture
    def sample_stencil(self):
        def _stencil(field_out: gtscript.Field[gtscript.IJ, np.float64]):
            with computation(FORWARD), interval(...):
                field_out = 1.0

        return gtscript.stencil(backend="numpy", definition=_stencil)

    def test_ndarray(self, sample_stencil):
        with pytest.raises(ValueError, match="Storage for '.*' has 3 dimensions but the API signature expects 2 .*"):
            sample_stencil(field_out=np.ndarray((3, 3, 3), np.float64))

    def test_storage(self, sample_stencil):
        with pytest.raises(Exception, match="Storage for '.*' has dimensions '.*' but the API signature expects '\\[I, J\\]'"):
            sample_stencil(field_out=DimensionsWrapper(
                array=gt_storage.empty(shape=(3, 3), dimensions=["I", "K"], dtype=np.float64, backend="numpy",
                                       aligned_index=(0, 0)),
                dimensions=("I", "K")))

    def test_ij_field_is_written(self, sample_stencil):
        f = gt_storage.zeros(shape=(3, 4), dimensions=["I", "J"], dtype=np.float64, backend="numpy")
        sample_stencil(field_out=f)
        assert (f == 1.0).all()


class TestDataDimensions:
    """test_call_interface.py:381-459."""

    @pytest.fixture
    def sample_stencil(self):
        def _stencil(field_out: gtscript.Field[gtscript.IJK, (np.float64, (2,))]):
            with computation(FORWARD), interval(...):
                field_out[0, 0, 0][0] = 0.0
                field_out[0, 0, 0][1] = 1.0

        return gtscript.stencil(backend="numpy", definition=_stencil)

    def test_mismatch(self, sample_stencil):
        with pytest.raises(ValueError, match="Field '.*' expects data dimensions \\(2,\\) but got \\(3,\\)"):
            sample_stencil(field_out=gt_storage.empty(shape=(3, 3, 1), dimensions=["I", "J", "K"], dtype=(np.float64, (3,)),
                                                      backend="numpy", aligned_index=(0, 0, 0)))

    @pytest.mark.parametrize("backend", ["numpy", pytest.param("hip:mi300", marks=pytest.mark.gpu)])
    def test_data_dimension_1d(self, backend):
        @gtscript.stencil(backend=backend)
        def data_dimension_1d(field_out: gtscript.Field[gtscript.IJ, (np.float64, (1,))]):
            with computation(FORWARD), interval(...):
                field_out[0, 0][0] = 42.0

        ones = gt_storage.ones(shape=(2, 3), dimensions=["I", "J"], dtype=(np.float64, (1,)), backend=backend,
                               aligned_index=(0, 0))
        data_dimension_1d(ones)
        assert (gt_storage.asnumpy(ones) == 42.0).all() and ones.shape == (2, 3, 1)

    def test_data_dimensions_1d_error(self):
        from gt4py_amd.cartesian.definitions import GTScriptSyntaxError

        with pytest.raises(GTScriptSyntaxError, match="Data index out of bounds."):

            @gtscript.stencil(backend="numpy")
            def data_dimension_1d_error(field_out: gtscript.Field[gtscript.IJ, (np.float64, (1,))]):
                with computation(FORWARD), interval(...):
                    field_out[0, 0][1] = 42.0


def calc_damp(outp: Field[float], inp: Field[K, float]):
    with computation(FORWARD), interval(...):
        outp = inp


def test_origin_unchanged():
    """test_call_interface.py:462-482: the caller's origin dict may gain keys, never lose/alter them."""
    stencil = gtscript.stencil(backend="numpy", definition=calc_damp)
    outp = gt_storage.ones(backend="numpy", aligned_index=(1, 1, 1), shape=(4, 4, 4), dtype=float, dimensions="IJK")
    inp = gt_storage.ones(backend="numpy", aligned_index=(1,), shape=(4,), dtype=float, dimensions="K")
    origin = {"_all_": (1, 1, 1), "inp": (1,)}
    origin_ref = copy.deepcopy(origin)
    stencil(outp, inp, origin=origin, domain=(3, 3, 3))
    assert all(origin.get(k) == v for k, v in origin_ref.items())


def test_permute_axes():
    """test_call_interface.py:485-503: __gt_dims__ triggers a transpose into I, J, K order."""
    stencil = gtscript.stencil(backend="numpy", definition=calc_damp)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        outp = gt_storage.ones(backend="numpy", aligned_index=(1, 1, 1), shape=(4, 4, 4), dtype=float, dimensions="KJI")
        inp = gt_storage.from_array(data=np.arange(4), backend="numpy", aligned_index=(1,), dtype=float, dimensions="K")
        stencil(DimensionsWrapper(array=outp, dimensions="KJI"), inp)
    for i in range(4):
        np.testing.assert_equal(outp[i, :, :], i)


def test_dtype_and_parameter_type_errors():
    stencil = gtscript.stencil(definition=base_stencil, backend="numpy")
    A = np.ones((3, 3, 3))
    with pytest.raises(TypeError, match="The dtype of field 'field3' is 'float64' instead of 'float32'"):
        stencil(A, A.copy(), A.copy(), param=3.0)
    C = np.ones((3, 3, 3), np.float32)
    with pytest.raises(TypeError, match="The type of parameter 'param'"):
        stencil(A, A.copy(), C, param=np.float32(3.0))
    with pytest.raises(ValueError, match="Compute domain too small|Invalid 'domain'"):
        stencil(A, A.copy(), C, param=3.0, domain=(1, 1))
    with pytest.raises(ValueError, match="Invalid 'origin' value"):
        stencil(A, A.copy(), C, param=3.0, origin="nonsense")


def test_domain_origin_cache_and_singleton():
    """test_stencil_object.py:25-60: one instance per class; (domain, origin) cached per call shape."""
    s1 = gtscript.stencil(definition=avg_stencil, backend="numpy")
    s2 = gtscript.stencil(definition=avg_stencil, backend="numpy")
    assert s1 is s2 and s1 == s2 and hash(s1) == hash(s2)
    with pytest.raises(AttributeError):
        s1.backend = "other"
    s1.clean_call_args_cache()
    a, b = np.ones((8, 8, 3)), np.zeros((8, 8, 3))
    s1(a, b, origin=(1, 1, 0))
    s1(a, b, origin=(1, 1, 0))
    assert len(type(s1)._domain_origin_cache) == 1
    s1(a, b, origin=(2, 2, 0), domain=(3, 3, 3))
    assert len(type(s1)._domain_origin_cache) == 2
    s1.clean_call_args_cache()
    assert len(type(s1)._domain_origin_cache) == 0
    assert s1.backend == "numpy" and "out_field" in s1.field_info and s1.options["name"] == "avg_stencil"
    assert "avg_stencil" in s1.source and s1.domain_info.parallel_axes == ("I", "J")


def test_freeze_skips_validation_and_runs():
    s = gtscript.stencil(definition=avg_stencil, backend="numpy")
    a, b = np.ones((8, 8, 3)), np.zeros((8, 8, 3))
    frozen = s.freeze(origin={"in_field": (1, 1, 0), "out_field": (1, 1, 0)}, domain=(6, 6, 3))
    frozen(in_field=a, out_field=b)
    assert (b[1:-1, 1:-1] == 1).all() and b.sum() == 6 * 6 * 3
    with pytest.raises(ValueError, match="origin"):
        s.freeze(origin={"in_field": (1, 1, 0)}, domain=(6, 6, 3))


def test_layout_warning():
    """test_field_layouts.py:51-80: a non-optimal layout only warns."""
    s = gtscript.stencil(definition=avg_stencil, backend="numpy")
    s.clean_call_args_cache()
    a = np.asfortranarray(np.ones((9, 9, 3)))
    b = np.zeros((9, 9, 3))
    with pytest.warns(UserWarning, match="layout of the field 'in_field' is not recommended"):
        s(a, b, origin=(1, 1, 0))
    assert (b[1:-1, 1:-1] == 1).all()


def test_backend_registry_feedback():
    """test_backend.py:166-171, test_stencil_object.py:71-78."""
    with pytest.raises(ValueError, match="Backend 'xla' is not registered. Valid options are:"):
        gtscript.stencil(definition=avg_stencil, backend="xla")
    assert "hip:mi300" in gt_backend.REGISTRY and gt_backend.from_name("hip:mi300").storage_info["device"] == "gpu"
    with pytest.warns(RuntimeWarning, match="Unknown options"):
        gtscript.stencil(definition=avg_stencil, backend="numpy", no_such_option=1, rebuild=True)


def test_numpy_plumbing_config0():
    """BASELINE config[0]: 5-point Laplacian 128x128x64 fp64 via @gtscript.stencil on backend=numpy."""
    from oracle import ref_numpy as R

    @gtscript.stencil(backend="numpy")
    def lap(inp: Field[np.float64], out: Field[np.float64]):
        with computation(PARALLEL), interval(...):
            out = -4.0 * inp[0, 0, 0] + inp[-1, 0, 0] + inp[1, 0, 0] + inp[0, -1, 0] + inp[0, 1, 0]

    rng = np.random.default_rng(1337)
    inp = gt_storage.from_array(rng.uniform(-1, 1, (130, 130, 64)), backend="numpy", aligned_index=(1, 1, 0))
    out = gt_storage.zeros((130, 130, 64), backend="numpy", aligned_index=(1, 1, 0))
    lap(inp, out, origin=(1, 1, 0), domain=(128, 128, 64))
    want = np.zeros((130, 130, 64))
    R.laplacian(np.asarray(inp), want)
    assert np.array_equal(out, want)


def test_generic_evaluator_equals_handwritten_restatements():
    """oracle/numpy_backend.py (shares the product frontend) == oracle/ref_numpy.py (independent)."""
    from gt4py_amd.cartesian.backend import hip_templates as T
    from oracle import ref_numpy as R

    rng = np.random.default_rng(8)
    for dt in (np.float64, np.float32):
        hd = gtscript.stencil(backend="numpy", definition=T.hdiff_limiter_field, dtypes={"T": dt}, rebuild=True)
        u = rng.uniform(-10, 10, (21, 17, 3)).astype(dt)
        c = rng.uniform(0, 0.5, (21, 17, 3)).astype(dt)
        got, want = np.zeros_like(u), np.zeros_like(u)
        hd(u, got, c, origin=(2, 2, 0))
        R.hdiff(u, want, c)
        assert np.array_equal(got, want)
    tri = gtscript.stencil(backend="numpy", definition=T.tridiagonal_solver, dtypes={"T": np.float64}, rebuild=True)
    shape = (6, 5, 17)
    inf, diag = rng.uniform(-1, 1, shape), rng.uniform(4, 5, shape)
    sup, rhs = rng.uniform(-1, 1, shape), rng.uniform(-10, 10, shape)
    s1, r1, o1 = sup.copy(), rhs.copy(), np.zeros(shape)
    s2, r2, o2 = sup.copy(), rhs.copy(), np.zeros(shape)
    tri(inf, diag, s1, r1, o1)
    R.tridiag(inf, diag, s2, r2, o2)
    assert np.array_equal(o1, o2) and np.array_equal(s1, s2) and np.array_equal(r1, r2)
    with pytest.raises(ValueError, match="Compute domain too small"):
        tri(inf[:, :, :1], diag[:, :, :1], s1[:, :, :1], r1[:, :, :1], o1[:, :, :1])


def test_every_call_validation_message():
    """The remaining branches of stencil_object.py:342-494 of the reference, message by message: a zero-sized
    domain, a domain the fields cannot serve (with the offending fields named), an origin inside the stencil's
    reach, a missing field and a missing parameter."""
    stencil = gtscript.stencil(definition=avg_stencil, backend="numpy")
    in_field, out_field = _in_out("numpy")
    with pytest.raises(ValueError, match="Compute domain contains zero sizes"):
        stencil(in_field=in_field, out_field=out_field, origin=(1, 1, 0), domain=(0, 0, 0))
    # ... which, by the partial order of Shape (gtc/definitions.py:141-171: "no element smaller, any element
    # larger"), only refuses the all-zero domain: one empty axis passes and the call writes nothing
    stencil(in_field=in_field, out_field=out_field, origin=(1, 1, 0), domain=(0, 4, 4))
    stencil(in_field=in_field, out_field=out_field, origin=(1, 1, 0), domain=(4, 4, 0))
    assert (out_field.array == 0).all()
    with pytest.raises(ValueError, match="Invalid 'domain' value"):
        stencil(in_field=in_field, out_field=out_field, origin=(1, 1, 0), domain=(4, 4))
    with pytest.raises(ValueError, match=r"Compute domain too large for stencil[\s\S]*Offending fields[\s\S]*in_field"):
        stencil(in_field=in_field, out_field=out_field, origin=(1, 1, 0), domain=(21, 20, 10))
    with pytest.raises(ValueError, match="Origin for field in_field too small. Must be at least"):
        stencil(in_field=in_field, out_field=out_field, origin={"in_field": (0, 1, 0), "out_field": (1, 1, 0)},
                domain=(4, 4, 4))
    with pytest.raises((ValueError, TypeError), match="in_field|out_field"):
        stencil(in_field=in_field, domain=(4, 4, 4), origin=(1, 1, 0))
    pstencil = gtscript.stencil(definition=base_stencil, backend="numpy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        A, B, C = _abc()
        with pytest.raises((ValueError, TypeError), match="param"):
            pstencil(A, B, C, domain=(1, 1, 1))
    # nothing was written by any of the refused calls
    assert (out_field.array == 0).all() and (A == 1).all()
