"""The reference's code-generation known-answer cases, run on the oracle (CPU) and on ``hip:mi300`` (GPU).

Each test restates one case of /root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/
test_code_generation.py (cited per test): a small GTScript definition, hand-made inputs and the values the
reference asserts.  Nothing here goes through the reference at run time.  Where the reference restricts a
feature to some of its backends the docstring says which; ``hip:mi300`` supports every case below.
"""

from enum import IntEnum

import numpy as np
import pytest

import gt4py_amd.storage as gt_storage
import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
from gt4py_amd.cartesian import gtscript
from gt4py_amd.cartesian.gtscript import (  # noqa: F401
    BACKWARD, FORWARD, IJ, IJK, PARALLEL, Field, GlobalTable, I, J, K, __INLINED, computation, horizontal, interval,
    region, sin, sqrt, tan, isfinite, isinf, isnan,
)

import interp_backend  # noqa: E402,F401 - registers the test-only backend "interp" (the independent interpreter behind the call interface)

BACKENDS = ["numpy", "interp", pytest.param("hip:mi300", marks=pytest.mark.gpu)]
F8 = np.float64


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


def on(backend, **options):
    """``@on(backend)``: build the definition below it for ``backend`` (``gtscript.stencil`` as a decorator)."""
    return gtscript.stencil(backend=backend, **options)


def host(a):
    return gt_storage.asnumpy(a) if not isinstance(a, np.ndarray) else a


class Alloc:
    """``gt_storage`` bound to one backend, with zero aligned_index unless given."""

    def __init__(self, backend):
        self.backend = backend

    def _kw(self, shape, kw):
        kw.setdefault("aligned_index", (0,) * len(shape))
        return dict(backend=self.backend, **kw)

    def zeros(self, shape, dtype=F8, **kw):
        return gt_storage.zeros(shape, dtype, **self._kw(shape, kw))

    def ones(self, shape, dtype=F8, **kw):
        return gt_storage.ones(shape, dtype, **self._kw(shape, kw))

    def full(self, shape, value, dtype=F8, **kw):
        return gt_storage.full(shape, value, dtype, **self._kw(shape, kw))

    def array(self, data, dtype=None, **kw):
        data = np.asarray(data)
        n_cartesian = len(kw.get("dimensions", "IJK"[: data.ndim]))
        kw.setdefault("aligned_index", (0,) * n_cartesian)
        return gt_storage.from_array(data, dtype=dtype or data.dtype, backend=self.backend, **kw)


@pytest.fixture
def mk(backend):
    return Alloc(backend)


# ---- :74-147 ----------------------------------------------------------------------------------------
def test_lazy_stencil(backend):
    @gtscript.lazy_stencil(backend=backend)
    def definition(phi_a: Field[F8], phi_b: Field[F8]):
        with computation(PARALLEL), interval(...):
            phi_a[0, 0, 0] = phi_b


def test_temporary_declared_in_if(backend, mk):
    @on(backend)
    def definition(phi_a: Field[F8]):
        with computation(PARALLEL), interval(...):
            if phi_a < 0:
                phi_b = -phi_a
            else:
                phi_b = phi_a
            phi_a = phi_b

    a = mk.array(np.linspace(-3, 3, 24).reshape(2, 3, 4))
    definition(a)
    np.testing.assert_array_equal(host(a), np.abs(np.linspace(-3, 3, 24).reshape(2, 3, 4)))


def test_stage_and_stencil_without_effect(backend, mk):
    @on(backend)
    def stage_only(phi_a: Field[F8]):
        with computation(PARALLEL), interval(...):
            field_c = 0.0  # noqa: F841

    def switched_off(f_in: Field[F8]):
        from __externals__ import flag

        with computation(PARALLEL), interval(...):
            if __INLINED(flag):
                lower = f_in  # noqa: F841

    off = gtscript.stencil(backend, switched_off, externals={"flag": False})
    f = mk.ones((23, 23, 23))
    stage_only(f, domain=(3, 3, 3))
    off(f, domain=(3, 3, 3))
    stage_only(f)
    assert (host(f) == 1).all()


# ---- :150-175 ---------------------------------------------------------------------------------------
def test_interval_blocks_keep_their_order(backend, mk):
    @on(backend)
    def stencil(u_old: Field[F8], u_new: Field[F8]):
        with computation(BACKWARD):
            with interval(-2, -1):
                u_new = u_old
            with interval(0, -2):
                u_new = u_old
        with computation(BACKWARD):
            with interval(-1, None):
                u_new = 2 * u_old
            with interval(0, -1):
                u_new[0, 0, 0] = 3 * u_old

    fin, fout = mk.ones((23, 23, 23)), mk.zeros((23, 23, 23))
    stencil(fin, fout)
    assert (host(fout)[:, :, :-1] == 3).all() and (host(fout)[:, :, -1] == 2).all()


# ---- :178-313 ---------------------------------------------------------------------------------------
def test_lower_dimensional_inputs(backend, mk):
    @on(backend)
    def stencil(field_3d: Field[IJK, F8], field_2d: Field[IJ, F8], field_1d: Field[K, F8]):
        with computation(PARALLEL):
            with interval(0, -1):
                work = field_2d + field_1d[1]
            with interval(-1, None):
                work = field_2d + field_1d[0]
        with computation(PARALLEL):
            with interval(0, 1):
                field_3d = work[1, 0, 0] + field_1d[1]
            with interval(1, None):
                field_3d[0, 0, 0] = work[-1, 0, 0]

    f3 = mk.zeros((6, 6, 6), aligned_index=(1, 1, 0))
    f2 = mk.zeros((6, 6), aligned_index=(1, 1), dimensions="IJ")
    f1 = mk.ones((6,), aligned_index=(0,), dimensions="K")
    assert f3.shape == (6, 6, 6) and f2.shape == (6, 6) and tuple(f1.shape) == (6,)
    stencil(f3, f2, f1, origin=(1, 1, 0), domain=(4, 3, 6))
    res = host(f3)
    assert (res[1:-1, 1:-2, :1] == 2).all() and (res[1:-1, 1:-2, 1:] == 1).all()
    stencil(f3, f2, f1, origin=(1, 1, 0))


def test_lower_dimensional_masked(backend, mk):
    @on(backend)
    def parallel(flagged: Field[IJK, F8], fed: Field[IJ, F8], result: Field[IJK, F8]):
        with computation(PARALLEL), interval(...):
            if flagged > 0.0:
                result[0, 0, 0] = fed

    @on(backend)
    def forward(flagged: Field[IJK, F8], fed: Field[IJ, F8], result: Field[IJK, F8]):
        with computation(FORWARD), interval(...):
            if flagged > 0.0:
                result[0, 0, 0] = fed

    rng = np.random.default_rng(1337)
    fed, result, flagged = rng.standard_normal((10, 10)), rng.standard_normal((10, 10, 10)), rng.standard_normal((10, 10, 10))
    for copy_2to3 in (parallel, forward):
        out_f = mk.array(result)
        copy_2to3(mk.array(flagged), mk.array(fed, dimensions="IJ"), out_f)
        np.testing.assert_array_equal(host(out_f), np.where(flagged > 0.0, fed[:, :, None], result))


def test_lower_dimensional_2d_to_3d_forward(backend, mk):
    @on(backend)
    def copy_2to3(fed: Field[IJ, F8], result: Field[IJK, F8]):
        with computation(FORWARD), interval(...):
            result[0, 0, 0] = fed

    rng = np.random.default_rng(7)
    fed = rng.standard_normal((10, 10))
    out_f = mk.array(rng.standard_normal((10, 10, 10)))
    copy_2to3(mk.array(fed, dimensions="IJ"), out_f)
    np.testing.assert_array_equal(host(out_f), np.broadcast_to(fed[:, :, None], (10, 10, 10)))


# ---- :316-368 ---------------------------------------------------------------------------------------
def test_higher_dimensional_fields(backend, mk):
    VEC2, MAT22 = (F8, (2,)), (F8, (2, 2))

    @on(backend)
    def stencil(phi: Field[F8], vector: Field[VEC2], mat_field: Field[MAT22]):
        with computation(PARALLEL), interval(...):
            work = vector[0, 0, 0][0] + vector[0, 0, 0][1]  # noqa: F841
        with computation(FORWARD):
            with interval(0, 1):
                vector[0, 0, 0][0] = phi[1, 0, 0]
                vector[0, 0, 0][1] = phi[0, 1, 0]
            with interval(1, -1):
                vector[0, 0, 0][0] = 2 * phi[1, 0, -1]
                vector[0, 0, 0][1] = 2 * phi[0, 1, -1]
            with interval(-1, None):
                vector[0, 0, 0][0] = phi[1, 0, 0]
                vector[0, 0, 0][1] = phi[0, 1, 0]
        with computation(PARALLEL), interval(...):
            mat_field[0, 0, 0][0, 0] = vector[0, 0, 0][0] + 1.0
            mat_field[0, 0, 0][1, 1] = vector[0, 0, 0][1] + 1.0

    phi = mk.ones((6, 6, 6), aligned_index=(1, 1, 0))
    vec = mk.full((6, 6, 6), 2.0, VEC2, aligned_index=(1, 1, 0))
    mat = mk.ones((6, 6, 6), MAT22, aligned_index=(1, 1, 0))
    assert vec.shape == (6, 6, 6, 2) and mat.shape == (6, 6, 6, 2, 2)
    stencil(phi, vec, mat, origin=(1, 1, 0), domain=(4, 4, 6))
    v, m = host(vec), host(mat)
    assert (v[1:-1, 1:-1, 0] == 1).all() and (v[1:-1, 1:-1, 1:-1] == 2).all() and (v[1:-1, 1:-1, -1] == 1).all()
    assert (m[1:-1, 1:-1, 1:-1, 0, 0] == 3).all() and (m[1:-1, 1:-1, 1:-1, 1, 1] == 3).all()
    assert (m[1:-1, 1:-1, :, 0, 1] == 1).all() and (m[0] == 1).all()
    stencil(phi, vec, mat)


def test_input_order(backend, mk):
    @on(backend)
    def stencil(source: Field[F8], parameter: F8, target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = source * parameter

    fout = mk.zeros((23, 23, 23))
    stencil(mk.ones((23, 23, 23)), 3.1415, fout)
    assert (host(fout) == 3.1415).all()


# ---- :394-446 ---------------------------------------------------------------------------------------
def test_variable_offsets(backend, mk):
    @on(backend)
    def stencil_ij(source: Field[F8], target: Field[F8], shift: Field[IJ, int]):
        with computation(FORWARD), interval(...):
            target[0, 0, 0] = source[0, 0, 1] + source[0, 0, shift + 1]
            shift = shift + 1

    @on(backend)
    def stencil_ijk(source: Field[F8], target: Field[F8], shift: Field[int]):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = source[0, 0, 1] + source[0, 0, shift + 1]

    data = np.arange(3 * 2 * 8, dtype=F8).reshape(3, 2, 8)
    out = mk.zeros((3, 2, 8))
    idx = mk.full((3, 2), -1, np.int64, dimensions="IJ")
    # level k reads in[k + 1] and in[k + (idx0 + k) + 1] with idx0 = -1 -> in[2k]; domain of 4 levels stays inside
    stencil_ij(mk.array(data), out, idx, domain=(3, 2, 4))
    k = np.arange(4)
    np.testing.assert_array_equal(host(out)[:, :, :4], data[:, :, k + 1] + data[:, :, 2 * k])
    assert (host(idx) == 3).all()
    out3 = mk.zeros((3, 2, 8))
    stencil_ijk(mk.array(data), out3, mk.full((3, 2, 8), -1, np.int64), domain=(3, 2, 7))
    np.testing.assert_array_equal(host(out3)[:, :, :7], data[:, :, 1:8] + data[:, :, 0:7])


def test_variable_offsets_and_while_loop(backend, mk):
    @on(backend)
    def stencil(edge_from: Field[F8], edge_to: Field[F8], q_from: Field[F8], q_to: Field[F8], lvl: Field[IJ, np.int_]):
        with computation(FORWARD), interval(0, -1):
            if edge_to[0, 0, 1] <= edge_from[0, 0, lvl]:
                q_to = q_from[0, 0, 1]
            else:
                qsum = edge_from[0, 0, lvl + 1] - edge_to[0, 0, lvl]
                while edge_from[0, 0, lvl + 1] < edge_to[0, 0, 1]:
                    qsum += q_from[0, 0, lvl] / (edge_to[0, 0, 1] - edge_from[0, 0, lvl])
                    lvl = lvl + 1
                q_to[0, 0, 0] = qsum / (edge_to[0, 0, 1] - edge_to)

    # edge_to[k+1] <= edge_from[k + lvl] everywhere: only the first branch runs
    shape = (2, 2, 5)
    edge_from, edge_to = mk.full(shape, 10.0), mk.full(shape, 1.0)
    q_from = mk.array(np.arange(20, dtype=F8).reshape(shape))
    q_to = mk.zeros(shape)
    stencil(edge_from, edge_to, q_from, q_to, mk.zeros((2, 2), np.int_, dimensions="IJ"))
    np.testing.assert_array_equal(host(q_to)[:, :, :-1], host(q_from)[:, :, 1:])


def test_nested_while_loop(backend, mk):
    @on(backend)
    def stencil(phi_a: Field[F8], phi_b: Field[np.int_]):
        with computation(PARALLEL), interval(...):
            while phi_a < 1:
                add = 0
                while phi_a + phi_b < 1:
                    add += 1
                phi_a += add

    # the inner loop never runs for b = 2; a >= 1 never enters
    a = mk.array(np.array([1.0, 2.0, 5.0, 7.0]).reshape(1, 1, 4))
    stencil(a, mk.full((1, 1, 4), 2, np.int_))
    np.testing.assert_array_equal(host(a).ravel(), [1.0, 2.0, 5.0, 7.0])


# ---- :449-517 ---------------------------------------------------------------------------------------
def test_mask_with_offset_written_in_conditional(backend, mk):
    @on(backend)
    def stencil(result: Field[F8]):
        with computation(PARALLEL), interval(...):
            flagged = True
            if flagged[0, -1, 0] or flagged[0, 0, 0]:
                result = 1.0
            else:
                result[0, 0, 0] = 0.0

    result = mk.zeros((10, 10, 10))
    stencil(result)
    assert (host(result) == 1.0).all()


def test_data_dim_indirect_addressing(backend, mk):
    VEC2 = (np.int32, (2,))

    @on(backend)
    def write(input_field: Field[IJK, np.int32], output_field: Field[IJK, VEC2], index: int):
        with computation(PARALLEL), interval(...):
            output_field[0, 0, 0][index] = input_field

    @on(backend)
    def read(input_field: Field[IJK, VEC2], output_field: Field[IJK, np.int32], index: int):
        with computation(PARALLEL), interval(...):
            output_field[0, 0, 0] = input_field[0, 0, 0][index]

    out = mk.zeros((1, 1, 2), VEC2)
    write(mk.ones((1, 1, 2), np.int32), out, 1)
    np.testing.assert_array_equal(host(out)[0, 0], [[0, 1], [0, 1]])
    src = mk.array(np.array([[[[3, 4], [5, 6]]]], dtype=np.int32), dimensions=("I", "J", "K", "0"))
    out = mk.zeros((1, 1, 2), np.int32)
    read(src, out, 1)
    np.testing.assert_array_equal(host(out)[0, 0], [4, 6])


# ---- :520-586 ---------------------------------------------------------------------------------------
def test_negative_origin(backend, mk):
    @on(backend)
    def stencil_i(input_field: Field[IJK, np.int32], output_field: Field[IJK, np.int32]):
        with computation(PARALLEL), interval(...):
            output_field[0, 0, 0] = input_field[1, 0, 0]

    @on(backend)
    def stencil_k(input_field: Field[IJK, np.int32], output_field: Field[IJK, np.int32]):
        with computation(PARALLEL), interval(...):
            output_field[0, 0, 0] = input_field[0, 0, 1]

    for stencil, origin in ((stencil_i, (-1, 0, 0)), (stencil_k, (0, 0, -1))):
        out = mk.zeros((1, 1, 1), np.int32)
        stencil(mk.ones((1, 1, 1), np.int32), out, origin={"input_field": origin})
        assert host(out)[0, 0, 0] == 1


def test_origin_k_fields(backend, mk):
    @gtscript.stencil(backend=backend, rebuild=True)
    def k_to_ijk(result: Field[F8], fed: Field[K, F8]):
        with computation(PARALLEL), interval(...):
            result[0, 0, 0] = fed

    data = np.arange(10, dtype=F8)
    fed, result = mk.array(data, dimensions="K"), mk.zeros((2, 2, 10))
    k_to_ijk(result, fed, origin={"result": (0, 0, 1), "fed": (2,)}, domain=(2, 2, 8))
    res = host(result)
    np.testing.assert_array_equal(host(fed), data)
    np.testing.assert_array_equal(res[:, :, 1:-1], np.broadcast_to(data[2:], (2, 2, 8)))
    assert (res[:, :, 0] == 0).all() and (res[:, :, -1] == 0).all()


# ---- :589-677 ---------------------------------------------------------------------------------------
def test_tmp_stencil(backend, mk):
    @on(backend)
    def stencil(u_old: Field[F8], u_new: Field[F8]):
        with computation(PARALLEL):
            with interval(...):
                work = u_old + 1
        with computation(PARALLEL):
            with interval(...):
                u_new[0, 0, 0] = work[-1, 0, 0] + work[1, 0, 0]

    fout = mk.zeros((6, 6, 6))
    stencil(mk.ones((6, 6, 6)), fout, origin=(1, 1, 0), domain=(4, 4, 6))
    res = host(fout)
    assert (res[1:-1, 1:-1, :] == 4).all()
    assert (res[0] == 0).all() and (res[-1] == 0).all() and (res[:, 0] == 0).all() and (res[:, -1] == 0).all()


def test_backward_stencil(backend, mk):
    @on(backend)
    def stencil(u_old: Field[F8], u_new: Field[F8]):
        with computation(BACKWARD):
            with interval(-1, None):
                u_old = 2
                u_new = u_old
            with interval(0, -1):
                u_old = u_old[0, 0, 1] + 1
                u_new[0, 0, 0] = u_old

    fout = mk.zeros((4, 4, 4))
    stencil(mk.ones((4, 4, 4)), fout)
    np.testing.assert_array_equal(host(fout), np.broadcast_to([5.0, 4.0, 3.0, 2.0], (4, 4, 4)))


def test_while_stencil(backend, mk):
    @on(backend)
    def stencil(u_old: Field[F8], u_new: Field[F8]):
        with computation(PARALLEL):
            with interval(...):
                while u_old < 10:
                    u_old += 1
                u_new[0, 0, 0] = u_old

    fout = mk.zeros((6, 6, 6))
    stencil(mk.ones((6, 6, 6)), fout)
    assert (host(fout) == 10).all()


# ---- :680-750 ---------------------------------------------------------------------------------------
def test_higher_dim_literal_and_scalar_index(backend, mk):
    VEC4 = (F8, (4,))

    @on(backend)
    def literal(vector: Field[VEC4], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = vector[0, 0, 0][2]

    @on(backend)
    def scalar(vector: Field[VEC4], target: Field[F8], scalar_argument: int):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = vector[0, 0, 0][scalar_argument]

    data = np.ones((6, 6, 6, 4))
    data[..., 2] = 5
    for call in (lambda v, o: literal(v, o), lambda v, o: scalar(v, o, 2)):
        out = mk.zeros((6, 6, 6))
        call(mk.array(data, dimensions=("I", "J", "K", "0")), out)
        assert (host(out) == 5).all()


def test_data_dims_declared_with_numpy_ints(backend, mk):
    @on(backend)
    def stencil(target: Field[IJK, np.int32], source: Field[IJK, (np.int32, (np.int32(3)))]):
        with computation(PARALLEL), interval(...):
            target = source.A[0]

    out = mk.zeros((2, 2, 4), np.int32)
    stencil(out, mk.ones((2, 2, 4), (np.int32, (np.int32(3)))))
    assert (host(out) == 1).all()


# ---- :753-951 ---------------------------------------------------------------------------------------
def test_native_function_call(backend, mk):
    @on(backend)
    def stencil(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = source[0, 0, 0] + sin(0.848062)

    out = mk.zeros((4, 4, 4))
    stencil(mk.ones((4, 4, 4)), out)
    np.testing.assert_allclose(host(out), 1.75, rtol=1e-6)
    assert (host(out) == 1.0 + np.sin(0.848062)).all()


def test_unary_ternary_and_mask(backend, mk):
    @on(backend)
    def unary(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = -source[0, 0, 0]

    @on(backend)
    def ternary(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = source[0, 0, 0] if source > 10 else source[0, 0, 0] + 1

    @on(backend)
    def mask(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            if source[0, 0, 0] > 0:
                target[0, 0, 0] = source
            else:
                target[0, 0, 0] = 1

    out = mk.zeros((4, 4, 4))
    unary(mk.ones((4, 4, 4)), out)
    assert (host(out) == -1).all()
    data = np.ones((4, 4, 4))
    data[0, 0, 1] = 20
    out = mk.zeros((4, 4, 4))
    ternary(mk.array(data), out)
    assert host(out)[0, 0, 1] == 20 and (host(out)[1:, 1:, 1] == 2).all()
    data[0, 0, 1] = -20
    out = mk.zeros((4, 4, 4))
    mask(mk.array(data), out)
    assert (host(out) > 0).all()


def test_k_offset_from_scalar_and_field(backend, mk):
    @on(backend)
    def by_scalar(source: Field[F8], target: Field[F8], scalar_value: int):
        with computation(PARALLEL), interval(1, None):
            target[0, 0, 0] = source[0, 0, scalar_value]

    @on(backend)
    def by_field(source: Field[F8], target: Field[F8], idx_field: Field[IJ, np.int64]):
        with computation(PARALLEL), interval(1, None):
            target[0, 0, 0] = source[0, 0, idx_field + 1]

    data = np.ones((4, 4, 4))
    data[:, :, 0] = 10
    out = mk.zeros((4, 4, 4))
    by_scalar(mk.array(data), out, -1)
    assert (host(out)[:, :, 1] == 10).all() and (host(out)[:, :, 2:] == 1).all() and (host(out)[:, :, 0] == 0).all()
    out = mk.zeros((4, 4, 4))
    by_field(mk.array(data), out, mk.full((4, 4), -2, np.int64, dimensions="IJ"))
    assert (host(out)[:, :, 1] == 10).all()


def test_k_only_and_table_access(backend, mk):
    @on(backend)
    def k_only(source: Field[K, F8], target: Field[F8]):
        with computation(PARALLEL):
            with interval(0, 1):
                target[0, 0, 0] = source[1]
            with interval(1, None):
                target[0, 0, 0] = source[-1]

    @on(backend)
    def table(table_view: GlobalTable[(F8, (4))], target: Field[F8]):
        with computation(PARALLEL):
            with interval(0, 1):
                target[0, 0, 0] = table_view.A[1]
            with interval(1, None):
                target[0, 0, 0] = table_view.A[2]

    out = mk.zeros((4, 4, 4))
    k_only(mk.array(np.array([2.0, 3.0, 4.0, 5.0]), dimensions="K"), out)
    np.testing.assert_array_equal(host(out)[1, 1], [3, 2, 3, 4])
    out = mk.zeros((4, 4, 4))
    table(gt_storage.from_array(np.array([2.0, 3.0, 4.0, 5.0]), dtype=F8, backend=backend, aligned_index=(0,)), out)
    np.testing.assert_array_equal(host(out)[1, 1], [3, 4, 4, 4])


def test_direct_datadims_index(backend, mk):
    VEC4 = (F8, (2, 2, 2, 2))

    @on(backend)
    def stencil(out: Field[F8], fed: GlobalTable[VEC4]):
        with computation(PARALLEL), interval(...):
            out[0, 0, 0] = fed.A[1, 0, 1, 0]

    data = np.ones((2, 2, 2, 2))
    data[1, 0, 1, 0] = 42
    out = mk.zeros((2, 2, 2))
    stencil(out, gt_storage.from_array(data, dtype=F8, backend=backend, dimensions=("0", "1", "2", "3")))
    assert (host(out) == 42).all()


def test_pruned_args_match(backend, mk):
    @on(backend)
    def stencil(out: Field[F8], fed: Field[F8]):
        with computation(PARALLEL), interval(...):
            out = 0.0
            with horizontal(region[I[0] - 1, J[0] - 1]):
                out[0, 0, 0] = fed

    out = mk.ones((2, 2, 2))
    stencil(out, mk.zeros((2, 2, 2)))
    assert (host(out) == 0).all()


# ---- :972-1131 (writes with an offset in K) -----------------------------------------------------------
KV = np.arange(40.0, 44.0)


def test_k_offset_write_simple(backend, mk):
    @on(backend)
    def simple(upper: Field[F8], lower: Field[F8]):
        with computation(FORWARD), interval(...):
            lower[0, 0, 1] = upper

    a, b = mk.array(KV.reshape(1, 1, 4)), mk.zeros((1, 1, 4))
    simple(a, b, domain=(1, 1, 3))
    np.testing.assert_array_equal(host(b).ravel(), [0, 40, 41, 42])


def test_k_offset_write_forward(backend, mk):
    @on(backend)
    def forward(upper: Field[F8], lower: Field[F8], scalar: F8):
        with computation(FORWARD), interval(1, None):
            upper[0, 0, -1] = scalar
            lower[0, 0, 0] = upper

    a, b = mk.array(KV.reshape(1, 1, 4)), mk.zeros((1, 1, 4))
    forward(a, b, 2.0)
    np.testing.assert_array_equal(host(a).ravel(), [2, 2, 2, 43])
    np.testing.assert_array_equal(host(b).ravel(), [0, 41, 42, 43])


def test_k_offset_write_backward(backend, mk):
    @on(backend)
    def backward(upper: Field[F8], lower: Field[F8], scalar: F8):
        with computation(BACKWARD), interval(-1, None):
            upper = scalar
        with computation(BACKWARD), interval(1, None):
            upper[0, 0, -1] = scalar
            lower[0, 0, 0] = upper

    a, b = mk.array(KV.reshape(1, 1, 4)), mk.zeros((1, 1, 4))
    backward(a, b, 2.0)
    np.testing.assert_array_equal(host(a).ravel(), [2, 2, 2, 2])
    np.testing.assert_array_equal(host(b).ravel(), [0, 2, 2, 2])


def test_k_offset_write_conditional(backend, mk):
    @on(backend)
    def column_physics(upper: Field[F8], lower: Field[F8], scalar: F8):
        with computation(BACKWARD), interval(1, -1):
            if upper > 0 and lower > 0:
                upper[0, 0, -1] = scalar
                lower[0, 0, 1] = upper
            lvl = 1
            while upper >= 0 and lower >= 0:
                upper[0, 0, lvl] = -1
                lower = -1
                lvl = lvl + 1

    a, b = mk.array(KV.reshape(1, 1, 4)), mk.ones((1, 1, 4))
    column_physics(a, b, 2.0)
    # the reference's hand unrolling (test_code_generation.py:1090-1128)
    np.testing.assert_array_equal(host(a).ravel(), [2, 2, -1, -1])
    np.testing.assert_array_equal(host(b).ravel(), [1, -1, 2, 42])


def test_variable_k_offset_write(backend, mk):
    """:1651-1669"""

    @on(backend)
    def stencil(source: Field[F8], shift: Field[IJ, np.int32], target: Field[F8]):
        with computation(FORWARD), interval(...):
            target[0, 0, shift - 1] = source

    rng = np.random.default_rng(3)
    data = rng.standard_normal((5, 5, 5))
    out = mk.zeros((5, 5, 5))
    stencil(mk.array(data), mk.ones((5, 5), np.int32, dimensions="IJ"), out)
    np.testing.assert_array_equal(host(out), data)


# ---- :1150-1260 -------------------------------------------------------------------------------------
def test_function_inline_in_while(backend, mk):
    @gtscript.function
    def add_42(v):
        return v + 42

    @on(backend)
    def stencil(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            count = 1
            while count < 10:
                sa = add_42(target)
                target = source + sa
                count = count + 1

    out = mk.ones((5, 5, 2))
    stencil(mk.ones((5, 5, 2)), out)
    assert (host(out) == 388.0).all()


def test_cast_in_index(backend, mk):
    @on(backend)
    def cast_in_index(source: Field[F8], i32: np.int32, i64: np.int64, target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target[0, 0, 0] = source[0, 0, i32 - i64]

    data = np.arange(8.0).reshape(1, 1, 8)
    out = mk.zeros((1, 1, 8))
    cast_in_index(mk.array(data), np.int32(3), np.int64(2), out, domain=(1, 1, 7))
    np.testing.assert_array_equal(host(out).ravel()[:7], np.arange(1.0, 8.0))


def test_read_after_write_stencil_builds(backend):
    @on(backend)
    def lagrangian_contributions(q: Field[F8], edge_from: Field[F8], edge_to: Field[F8], q4_1: Field[F8], q4_2: Field[F8],
                                 q4_3: Field[F8], q4_4: Field[F8], dp1: Field[F8], lvl: Field[IJ, np.int64]):
        with computation(FORWARD), interval(...):
            pl = (edge_to - edge_from[0, 0, lvl]) / dp1[0, 0, lvl]
            if edge_to[0, 0, 1] <= edge_from[0, 0, lvl + 1]:
                pr = (edge_to[0, 0, 1] - edge_from[0, 0, lvl]) / dp1[0, 0, lvl]
                q[0, 0, 0] = (q4_2[0, 0, lvl] + 0.5 * (q4_4[0, 0, lvl] + q4_3[0, 0, lvl] - q4_2[0, 0, lvl]) * (pr + pl)
                              - q4_4[0, 0, lvl] * 1.0 / 3.0 * (pr * (pr + pl) + pl * pl))
            else:
                qsum = (edge_from[0, 0, lvl + 1] - edge_to) * (
                    q4_2[0, 0, lvl] + 0.5 * (q4_4[0, 0, lvl] + q4_3[0, 0, lvl] - q4_2[0, 0, lvl]) * (1.0 + pl)
                    - q4_4[0, 0, lvl] * 1.0 / 3.0 * (1.0 + pl * (1.0 + pl)))
                lvl = lvl + 1
                while edge_from[0, 0, lvl + 1] < edge_to[0, 0, 1]:
                    qsum += dp1[0, 0, lvl] * q4_1[0, 0, lvl]
                    lvl = lvl + 1
                dp = edge_to[0, 0, 1] - edge_from[0, 0, lvl]
                esl = dp / dp1[0, 0, lvl]
                qsum += dp * (q4_2[0, 0, lvl] + 0.5 * esl * (q4_3[0, 0, lvl] - q4_2[0, 0, lvl]
                                                             + q4_4[0, 0, lvl] * (1.0 - (2.0 / 3.0) * esl)))
                q = qsum / (edge_to[0, 0, 1] - edge_to)
            lvl = lvl - 1


# ---- :1278-1391 (absolute K index; reference: debug and dace backends only) ---------------------------
def test_absolute_k_index(backend, mk):
    @on(backend)
    def literal(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target = source.at(K=2)

    @on(backend)
    def parameter(source: Field[F8], target: Field[F8], idx: int):
        with computation(PARALLEL), interval(...):
            target = source.at(K=idx)

    @gtscript.stencil(backend=backend, externals={"K4": 4})
    def external(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            from __externals__ import K4

            target = source.at(K=K4)

    @on(backend)
    def from_field(source: Field[F8], shift: Field[IJ, np.int64], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target = source.at(K=shift)

    @on(backend)
    def computed(source: Field[F8], shift: Field[IJ, np.int32], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target = source.at(K=shift - 1)

    @on(backend)
    def lower_dim(k_field: Field[K, F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            target = k_field.at(K=2)

    @on(backend)
    def conditional(source: Field[F8], target: Field[F8]):
        with computation(PARALLEL), interval(...):
            k_level = 0
            while source.at(K=k_level) < 2:
                k_level += 1
            target[0, 0, 0] = k_level

    def marked(level, value=42.42):
        data = np.ones((5, 5, 5))
        data[:, :, level] = value
        return mk.array(data)

    def run(call):
        out = mk.zeros((5, 5, 5))
        call(out)
        return host(out)

    assert (run(lambda o: literal(marked(2), o)) == 42.42).all()
    assert (run(lambda o: parameter(marked(3), o, 3)) == 42.42).all()
    assert (run(lambda o: external(marked(4), o)) == 42.42).all()
    assert (run(lambda o: from_field(marked(1), mk.ones((5, 5), np.int64, dimensions="IJ"), o)) == 42.42).all()
    assert (run(lambda o: computed(marked(1), mk.full((5, 5), 2, np.int32, dimensions="IJ"), o)) == 42.42).all()
    k_data = np.zeros(5)
    k_data[2] = 42.42
    assert (run(lambda o: lower_dim(mk.array(k_data, dimensions="K"), o)) == 42.42).all()
    assert (run(lambda o: conditional(marked(3, 10.0), o)) == 3).all()


# ---- :1394-1428 (K as a value; reference: debug, numpy, dace:cpu) --------------------------------------
def test_iterator_access(backend, mk):
    @on(backend)
    def stencil(field_A: Field[F8], field_B: Field[F8], offsets: Field[K, np.int32]):
        with computation(PARALLEL), interval(...):
            if K == 2:
                field_A = 20.20
            field_B = float(K + offsets)

    a, b = mk.zeros((3, 4, 5)), mk.zeros((3, 4, 5))
    stencil(a, b, mk.zeros((5,), np.int32, dimensions="K"))
    assert (host(a)[:, :, 2] == 20.20).all() and (np.delete(host(a), 2, axis=2) == 0).all()
    np.testing.assert_array_equal(host(b), np.broadcast_to(np.arange(5.0), (3, 4, 5)))


# ---- :1582-1637 (2-d temporaries; reference: debug, numpy, dace) ---------------------------------------
def test_2d_temporaries(backend, mk):
    @on(backend)
    def plain(source: Field[F8], target: Field[F8]):
        with computation(FORWARD), interval(0, 1):
            sheet: Field[IJ, F8] = 0
        with computation(FORWARD), interval(...):
            sheet = sheet + source
        with computation(FORWARD), interval(...):
            target = sheet

    out = mk.zeros((5, 5, 3))
    plain(mk.ones((5, 5, 3)), out)
    assert (host(out) == 3).all()

    @gtscript.stencil(backend=backend, dtypes={"MyFancySymbol": Field[IJ, F8]})
    def user_dtype(source: Field[F8], target: Field[F8]):
        with computation(FORWARD), interval(0, 1):
            sheet: MyFancySymbol = 0  # noqa: F821
        with computation(FORWARD), interval(...):
            target = sheet

    out = mk.ones((5, 5, 3))
    user_dtype(mk.ones((5, 5, 3)), out)
    assert (host(out) == 0).all()

    from gt4py_amd.cartesian.definitions import GTScriptSyntaxError

    with pytest.raises(GTScriptSyntaxError, match="Typed temporaries must be IJ,"):

        @on(backend)
        def k_temporary(source: Field[F8], target: Field[F8]):
            with computation(FORWARD), interval(0, 1):
                sheet: Field[K, F8] = 0
            with computation(FORWARD), interval(...):
                target = sheet


# ---- :1672-1690 -------------------------------------------------------------------------------------
def test_integer_power_arguments_are_left_alone(backend, mk):
    @on(backend)
    def stencil(source: Field[np.float32], squared: Field[IJ, np.int32], target: Field[np.float32]):
        with computation(FORWARD), interval(...):
            target = source**squared

    data = np.linspace(0.5, 3, 125, dtype=np.float32).reshape(5, 5, 5)
    out = mk.zeros((5, 5, 5), np.float32)
    stencil(mk.array(data), mk.full((5, 5), 2, np.int32, dimensions="IJ"), out)
    np.testing.assert_array_equal(host(out), data * data)


# ---- :1693-1785 (illegal and legal self-assignments) ---------------------------------------------------
def test_no_write_and_read_with_horizontal_offset(backend):
    with pytest.raises(ValueError, match="Self-assignment with offset in I or J is illegal."):

        @on(backend)
        def direct(phi: Field[F8]):
            with computation(PARALLEL), interval(...):
                phi = (phi[I - 1] + phi[I + 1]) / 2

    with pytest.raises(ValueError, match="Illegal write and read with horizontal offset"):

        @on(backend)
        def through_temporary(phi: Field[F8]):
            with computation(PARALLEL), interval(...):
                work = (phi[J - 1] + phi[J + 1]) / 2
                phi = work * 2


def test_k_offsets_in_parallel_loops(backend):
    with pytest.raises(ValueError, match="write and read with k-offsets in PARALLEL"):

        @on(backend)
        def direct(phi: Field[np.int32]):
            with computation(PARALLEL), interval(1, None):
                phi = phi[K - 1] * 2

    with pytest.raises(ValueError, match="write and read with k-offsets in PARALLEL"):

        @on(backend)
        def through_temporary(phi: Field[np.int32]):
            with computation(PARALLEL), interval(1, None):
                work = phi[K - 1]
                phi = work * 2

    with pytest.raises(ValueError, match="write and read with `VariableKOffset` and/or `AbsoluteKIndex`"):

        @on(backend)
        def absolute(phi: Field[np.int32]):
            with computation(PARALLEL), interval(...):
                level = phi.at(K=1)
                phi = 2 * level

    with pytest.raises(ValueError, match="write and read with `VariableKOffset` and/or `AbsoluteKIndex`"):

        @on(backend)
        def variable(phi: Field[np.int32], offset: int = -1):
            with computation(PARALLEL), interval(1, None):
                bottom = phi[0, 0, offset]
                phi = phi + 2 * bottom

    @on(backend)
    def center_read(phi: Field[np.int32]):
        with computation(PARALLEL), interval(...):
            phi = phi[0, 0, 0] * 2

    @on(backend)
    def center_write(phi: Field[np.int32]):
        with computation(PARALLEL), interval(...):
            phi[0, 0, 0] = phi * 2

    @on(backend)
    def index_fields(phi: Field[np.float32], index: Field[np.int32]):
        with computation(PARALLEL), interval(1, None):
            phi = index + index[K - 1] * 2

    @on(backend)
    def single_level_intervals(phi: Field[np.bool_]):
        with computation(PARALLEL):
            with interval(0, 1):
                phi = phi[K + 1]
            with interval(-1, None):
                phi = phi[K - 1]


def test_self_assignment_in_forward(backend, mk):
    @on(backend)
    def direct(phi: Field[np.int32]):
        with computation(FORWARD), interval(1, None):
            phi = phi[K - 1] * 2

    @on(backend)
    def through_temporary(phi: Field[np.int32]):
        with computation(FORWARD), interval(1, None):
            work = phi[K - 1]
            phi = work * 2

    for stencil in (direct, through_temporary):
        f = mk.ones((2, 2, 5), np.int32)
        stencil(f)
        np.testing.assert_array_equal(host(f), np.broadcast_to([1, 2, 4, 8, 16], (2, 2, 5)))


def test_reset_mask_2d(backend, mk):
    @on(backend)
    def stencil(dp1: Field[F8], edge_from: Field[F8], lvl: Field[IJ, np.int32]):
        with computation(PARALLEL), interval(0, -1):
            dp1 = edge_from[0, 0, 1] - edge_from
        with computation(FORWARD), interval(0, 1):
            lvl = 0

    mask = mk.ones((5, 5), np.int32, dimensions="IJ")
    stencil(mk.zeros((5, 5, 5)), mk.ones((5, 5, 5)), mask)
    assert (host(mask) == 0).all()


# ---- :1830-1857 (reference: debug, dace, gt:gpu) -------------------------------------------------------
def test_offset_j_in_temporaries(backend, mk):
    @gtscript.function
    def a_gtscript_function(b):
        return sqrt(abs(b[0, 1, 0]))

    @on(backend)
    def stencil(u_old: Field[IJK, F8], u_new: Field[IJK, F8]):
        with computation(PARALLEL), interval(...):
            abs_res = abs(u_old)
            tan_res = tan(abs_res)
            sqrt_res = a_gtscript_function(tan_res)
            u_new = (sqrt_res if isfinite(sqrt_res) else u_old if isinf(sqrt_res) else u_new
                         if isnan(sqrt_res) else 0.0)

    rng = np.random.default_rng(11)
    data = rng.uniform(-1.5, 1.5, (4, 6, 3))
    prev = rng.uniform(5, 6, (4, 5, 3))
    out = mk.array(prev)
    stencil(mk.array(data, aligned_index=(0, 0, 0)), out, domain=(4, 5, 3))
    with np.errstate(invalid="ignore"):
        s = np.sqrt(np.abs(np.tan(np.abs(data[:, 1:, :]))))
    want = np.where(np.isfinite(s), s, np.where(np.isinf(s), data[:, :5, :], np.where(np.isnan(s), prev, 0.0)))
    np.testing.assert_allclose(host(out), want, rtol=1e-13, atol=0)


# ---- :1860-1892 -------------------------------------------------------------------------------------
class MyEnum(IntEnum):
    Zero = 0
    Ten = 10
    Twenty = 20
    Thirty = 30


gtscript.enum(MyEnum)


def test_enum_runtime(backend, mk):
    @on(backend)
    def stencil(target: Field[int], order: MyEnum):
        with computation(PARALLEL), interval(0, 1):
            target = 32
            if order < MyEnum.Ten:
                target = MyEnum.Ten
        with computation(PARALLEL), interval(1, 2):
            target = 23
            target = MyEnum.Twenty
        with computation(PARALLEL), interval(2, None):
            target = 56
            target = MyEnum.Thirty

    out = mk.zeros((5, 5, 5), int)
    stencil(out, MyEnum.Zero)
    np.testing.assert_array_equal(host(out), np.broadcast_to([10, 20, 30, 30, 30], (5, 5, 5)))


# ---- test_math_functions.py:31-113 ----------------------------------------------------------------------
def test_math_functions(backend, mk):
    from gt4py_amd.cartesian.gtscript import erf, erfc, round, round_away_from_zero  # noqa: A004
    import scipy.special

    @on(backend)
    def stencil(x: Field[F8], out_erf: Field[F8], out_erfc: Field[F8], out_round: Field[F8], out_away: Field[F8]):
        with computation(PARALLEL), interval(...):
            out_erf = erf(x)
            out_erfc = erfc(x)
            out_round = round(x)
            out_away = round_away_from_zero(x)

    data = np.array([-2.5, -1.5, -0.5, -0.2, 0.0, 0.3, 0.5, 1.5, 2.5, 3.49, 5.5, 7.2]).reshape(2, 2, 3)
    outs = [mk.zeros((2, 2, 3)) for _ in range(4)]
    stencil(mk.array(data), *outs)
    np.testing.assert_allclose(host(outs[0]), scipy.special.erf(data), rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(host(outs[1]), scipy.special.erfc(data), rtol=1e-14, atol=1e-16)
    np.testing.assert_array_equal(host(outs[2]), np.round(data))  # half to even
    np.testing.assert_array_equal(host(outs[3]), np.trunc(data + np.copysign(0.5, data)))
