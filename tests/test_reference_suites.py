"""The reference's integration suites (tests/reference_suites.py) on both backends.

CPU (`-m "not gpu"`): backend "numpy" = oracle/numpy_backend.py -- this pins the oracle (and the shared
frontend: gtscript functions, run-time ifs, externals, optional fields, K offsets) to the reference's
own plain-numpy validations.  GPU (`-m gpu`): the same cases on backend "hip:mi300".
Comparison is exact: the validations perform the same IEEE operations in the same order.
"""

import numpy as np
import pytest

import reference_suites as rs

def _case_id(name, ext, domain):
    tags = "_".join("{}={}".format(k, getattr(v, "__name__", v)) for k, v in ext.items()) or "noext"
    return "{}-{}-{}".format(name, tags, "x".join(map(str, domain)))


CASES = [pytest.param(n, e, d, id=_case_id(n, e, d)) for n, e, d in rs.cases()]


def _run(backend, name, ext, domain, to_device=None):
    from gt4py_amd.cartesian import gtscript

    suite = rs.SUITES[name]
    arrays, origins, params, expected = rs.make_case(name, ext, domain)
    stencil = gtscript.stencil(backend=backend, definition=suite.definition, externals=ext)
    call_args = {}
    for fname, arr in arrays.items():
        switch = suite.optional.get(fname)
        if switch is not None and not ext[switch]:
            call_args[fname] = None
            origins.pop(fname)
            continue
        call_args[fname] = to_device(arr, origins[fname], suite.axes.get(fname, "IJK")) if to_device else arr.copy()
    stencil(**call_args, **params, origin=origins, domain=domain)
    for fname, want in expected.items():
        got = call_args[fname]
        got = got.get() if hasattr(got, "get") else np.asarray(got)
        boundary = [b for b, ax in zip(suite.fields[fname][1], "IJK") if ax in suite.axes.get(fname, "IJK")]
        np.testing.assert_array_equal(rs._inner(got, boundary), want.astype(got.dtype), err_msg=f"{name}: {fname}")
    # nothing but the declared outputs changed
    for fname, arr in arrays.items():
        if fname not in expected and call_args[fname] is not None:
            got = call_args[fname]
            got = got.get() if hasattr(got, "get") else np.asarray(got)
            np.testing.assert_array_equal(got, arr, err_msg=f"{name}: input {fname} was modified")
    return stencil


@pytest.mark.parametrize("name,ext,domain", CASES)
def test_oracle_backend_matches_reference_validation(name, ext, domain):
    import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"

    _run("numpy", name, ext, domain)


@pytest.mark.gpu
@pytest.mark.parametrize("name,ext,domain", CASES)
def test_hip_backend_matches_reference_validation(name, ext, domain):
    import gt4py_amd.storage as gt_storage

    def to_device(arr, origin, axes):
        n_data = arr.ndim - len(axes)
        return gt_storage.from_array(arr, dtype=arr.dtype, backend="hip:mi300", aligned_index=origin,
                                     dimensions=tuple(axes) + tuple(str(n) for n in range(n_data)))

    _run("hip:mi300", name, ext, domain, to_device)


def test_field_info_of_the_suites_matches_the_declared_boundaries():
    """The reference harness asserts that the analysed boundary equals the suite's declared one
    (gt4py/cartesian/testing/suites.py: `_test_generation`)."""
    import oracle.numpy_backend  # noqa: F401
    from gt4py_amd.cartesian import gtscript

    for name, suite in rs.SUITES.items():
        for ext in suite.externals:
            st = gtscript.stencil(backend="numpy", definition=suite.definition, externals=ext)
            for fname, (dt, boundary, _) in suite.fields.items():
                info = st.field_info[fname]
                switch = suite.optional.get(fname)
                if switch is not None and not ext[switch]:
                    assert info is None or info.access.name == "NONE", (name, fname)
                    continue
                present = suite.axes.get(fname, "IJK")
                got = tuple(tuple(max(0, v) for v in b) for b, ax in zip(info.boundary, "IJK") if ax in present)
                assert got == tuple(b for b, ax in zip(boundary, "IJK") if ax in present), (name, fname)
                assert info.dtype == np.dtype(dt) and tuple(info.data_dims) == tuple(suite.data_dims.get(fname, ()))
