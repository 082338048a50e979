"""The INDEPENDENT oracle (oracle/gtscript_interp.py): an interpreter of GTScript source that shares no code with the product --
no frontend, no IR, no analysis -- and restates the reference's rules from the reference's own source.

Why it exists: `oracle/numpy_backend.py` interprets the product's parsed IR, so a parse / extent / dtype-rule error would be common to
the product and its oracle ("What's weak" 4).  Here the chain is closed three ways:

* the interpreter is PINNED on what the reference's tests hold: the known-answer cases of its extent pass
  (/root/reference/tests/cartesian_tests/unit_tests/test_gtc/test_passes/test_oir_optimizations/test_utils.py:93-232) and the plain-numpy
  `validation` functions of its integration suites (tests/reference_suites.py, transcribed from
  /root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/test_suites.py);
* CPU: product frontend + analysis + numpy oracle against the interpreter, bit for bit, on the stencil zoo and on random programs of
  the three fuzz generators;
* GPU: `hip:mi300` (frontend + analysis + planner + generated HIP) against the interpreter directly -- no product code on the
  checking side at all.
"""

import os
import pathlib

import numpy as np
import pytest

import fuzz_stencils
import reference_suites as rs
import stencil_zoo as zoo
from oracle import gtscript_interp as gi

B = gi._Bound


def _at_endpt(end, start_offset, end_offset=None):
    """common.HorizontalInterval.at_endpt (gtc/common.py:822-831)"""
    return B(end, start_offset), B(end, start_offset + 1 if end_offset is None else end_offset)


def _compute_domain(start_offset=0, end_offset=0):
    return B(False, start_offset), B(True, end_offset)


# ---- pinned on the reference's own known-answer tests ------------------------------------------------------------------------------
def test_overlap_along_axis_known_answers():
    """test_utils.py:114-157, case by case."""
    f = gi.Interpreter._overlap_along_axis
    assert f((0, 0), _compute_domain()) == (0, 0)
    assert f((0, 0), _compute_domain(-1, 1)) == (0, 0)
    lo, hi = f((0, 0), _at_endpt(False, 2))
    assert lo == -2 and hi > 100
    assert f((0, 0), _at_endpt(False, -4)) is None
    assert f((0, 0), _at_endpt(True, 4)) is None
    lo, hi = f((-1, 1), _at_endpt(False, -4, 4))
    assert lo == 0 and hi > 100
    lo, hi = f((-1, 1), _at_endpt(True, -4, 4))
    assert lo < -100 and hi == 0


@pytest.mark.parametrize("region_i,offset,want", [
    (_at_endpt(True, 1), 1, ((0, 2), (0, 0))),
    (_at_endpt(True, 1), -1, ((0, 0), (0, 0))),
    (_at_endpt(True, 2), 0, None),
    ((None, None), -1, ((-1, 0), (0, 0))),
])
def test_access_extent_under_a_region_known_answers(region_i, offset, want):
    """test_utils.py:160-232: a read at I offset `offset` under a mask, in a block of extent ((0, 1), (0, 0))."""
    it = gi.Interpreter.__new__(gi.Interpreter)
    assert it._access_extent(((0, 1), (0, 0)), (offset, 0), (region_i, (None, None))) == want


def _simple(inp: "Field[np.float64]", output: "Field[np.float64]"):  # noqa: F821
    with computation(PARALLEL), interval(...):  # noqa: F821
        tmp = inp[1, 0, 0]
        output = tmp[1, 0, 0]  # noqa: F841


def test_stencil_extents_simple_known_answer():
    """test_utils.py:93-111: tmp = input[1]; output = tmp[1] -> input needs (1, 2), the first block runs on (0, 1)."""
    it = gi.Interpreter(_simple)
    inp, out = np.arange(5.0 * 2 * 1).reshape(5, 2, 1), np.zeros((3, 2, 1))
    it({"inp": (inp, (0, 0, 0)), "output": (out, (0, 0, 0))}, {}, (3, 2, 1))
    assert it.field_extent["inp"] == ((1, 2), (0, 0)) and it.field_extent.get("output", ((0, 0), (0, 0))) == ((0, 0), (0, 0))
    assert it.block_extent == [((0, 1), (0, 0)), ((0, 0), (0, 0))]
    np.testing.assert_array_equal(out, inp[2:5])


def test_upcasting_rule_examples():
    """gtir_upcaster.py:41-68 on the ranking of gtc/common.py:105-118: the lowest-ranking ufunc loop every operand fits."""
    f32, f64, i32, i64, b = (np.dtype(t) for t in (np.float32, np.float64, np.int32, np.int64, np.bool_))
    assert gi._ufunc_targets(np.add, [i64, f32]) == [f32, f32]      # not numpy's float64
    assert gi._ufunc_targets(np.add, [f32, f64]) == [f64, f64]
    assert gi._ufunc_targets(np.multiply, [i32, i64]) == [i64, i64]
    assert gi._ufunc_targets(np.true_divide, [i64, i64]) == [f32, f32]  # the first float loop both operands fit
    assert gi._ufunc_targets(np.greater, [f32, f64]) == [f64, f64]
    assert gi._ufunc_targets(np.logical_and, [b, b]) == [b, b]
    assert gi._ufunc_targets(np.sqrt, [f32]) == [f32]
    assert gi._ufunc_targets(np.maximum, [f32, i64]) == [f32, f32]


# ---- pinned on the reference suites' own validation functions ---------------------------------------------------------------------
SUITE_CASES = [pytest.param(n, e, d, id="{}-{}-{}".format(n, "_".join(f"{k}={getattr(v, '__name__', v)}" for k, v in e.items()) or "noext",
                                                          "x".join(map(str, d)))) for n, e, d in rs.cases()]


@pytest.mark.parametrize("name,ext,domain", SUITE_CASES)
def test_interpreter_reproduces_the_reference_suites_validations(name, ext, domain):
    suite = rs.SUITES[name]
    arrays, origins, params, expected = rs.make_case(name, ext, domain)
    got = {k: v.copy() for k, v in arrays.items() if not (suite.optional.get(k) is not None and not ext[suite.optional[k]])}
    gi.run(suite.definition, {k: (got[k], origins[k]) for k in got}, params, domain, externals=ext)  # (all 113 cases are restated)
    for fname, want in expected.items():
        boundary = [b for b, ax in zip(suite.fields[fname][1], "IJK") if ax in suite.axes.get(fname, "IJK")]
        np.testing.assert_array_equal(rs._inner(got[fname], boundary), want.astype(got[fname].dtype), err_msg=f"{name}: {fname}")
    for fname, arr in arrays.items():
        if fname not in expected and fname in got:
            np.testing.assert_array_equal(got[fname], arr, err_msg=f"{name}: input {fname} was modified")


# ---- product frontend + analysis + numpy oracle against the interpreter (CPU) ------------------------------------------------------
def _product_vs_interpreter(defn, externals, scalars, domain, seed, text=""):
    import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
    from gt4py_amd.cartesian import gtscript

    ref = gtscript.stencil(backend="numpy", definition=defn, externals=externals or None)
    ni, nj, nk = domain
    if "region[" in text:
        ni, nj = max(ni, 6), max(nj, 6)
    domain = (ni, nj, max(nk, ref.domain_info.min_sequential_axis_size))
    arrays, origins = zoo.make_inputs(ref, domain, seed)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, **scalars, origin=origins, domain=domain)
    got = {k: v.copy() for k, v in arrays.items()}
    it = gi.run(defn, {k: (got[k], origins[k]) for k in got}, scalars, domain, externals=externals)
    for k in arrays:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"field {k} (seed {seed})\n{text}")
    # what the product analysed must be what the interpreter analysed, independently: the extent of every API field
    for k, info in ref.field_info.items():
        if info is None or k not in it.field_extent:
            continue
        (ilo, ihi), (jlo, jhi) = it.field_extent[k]
        want = {"I": (max(0, -ilo), max(0, ihi)), "J": (max(0, -jlo), max(0, jhi))}
        for ax in "IJ":
            if ax in info.axes:
                assert tuple(max(0, int(b)) for b in info.boundary["IJK".index(ax)]) == want[ax], (k, ax, info.boundary, it.field_extent[k], text)
    return it


DOMAINS = [(9, 7, 5), (66, 5, 4), (3, 3, 2)]


@pytest.mark.parametrize("seed", range(int(os.environ.get("GT4MI_FUZZ_FIRST_SEED", "0")), int(os.environ.get("GT4MI_FUZZ_FIRST_SEED", "0")) + (int(os.environ.get("GT4MI_FUZZ_SEEDS", "0")) or 150)))
def test_product_frontend_agrees_with_the_interpreter_on_random_programs(seed, tmp_path):
    defn, scalars, text = fuzz_stencils.make_stencil(seed, tmp_path)
    _product_vs_interpreter(defn, {}, scalars, DOMAINS[seed % len(DOMAINS)], seed, text)


@pytest.mark.parametrize("seed", range(40))
def test_product_frontend_agrees_with_the_interpreter_on_two_sweep_programs(seed, tmp_path):
    defn, scalars, text = fuzz_stencils.make_two_sweep_stencil(seed, tmp_path)
    _product_vs_interpreter(defn, {}, scalars, (9, 7, 12), seed, text)


@pytest.mark.parametrize("seed", range(40))
def test_product_frontend_agrees_with_the_interpreter_on_shared_temporaries_programs(seed, tmp_path):
    defn, scalars, text = fuzz_stencils.make_shared_temporaries_stencil(seed, tmp_path)
    _product_vs_interpreter(defn, {}, scalars, (12, 9, 4), seed, text)


@pytest.mark.parametrize("name", sorted(zoo.ZOO))
def test_product_frontend_agrees_with_the_interpreter_on_the_zoo(name):
    defn, externals, scalars, _ = zoo.ZOO[name]
    _product_vs_interpreter(defn, externals, scalars, (9, 7, 12), 5)


def test_where_the_reference_backends_disagree_with_each_other(tmp_path):
    """Two places the interpreter's docstring names.  (1) `while`: the numpy backend re-evaluates the condition per body statement, the
    compiled backends loop per point; a program whose body falsifies the condition before its last statement gets different values.
    (2) a later interval block reading, at a horizontal offset, a temporary an earlier block of the same merged loop wrote: the
    reference's extent pass leaves the earlier write a zero extent.  Both are COUNTED over the fuzz corpus so that the numbers in
    DESIGN.md are reproducible, and (1) is shown on one program."""
    n_while = n_quirk = n_differ = 0
    for seed in range(150):
        defn, scalars, text = fuzz_stencils.make_stencil(seed, tmp_path)
        import oracle.numpy_backend  # noqa: F401
        from gt4py_amd.cartesian import gtscript

        ref = gtscript.stencil(backend="numpy", definition=defn)
        ni, nj, nk = DOMAINS[seed % 3]
        if "region[" in text:
            ni, nj = max(ni, 6), max(nj, 6)
        domain = (ni, nj, max(nk, ref.domain_info.min_sequential_axis_size))
        arrays, origins = zoo.make_inputs(ref, domain, seed)
        a = {k: v.copy() for k, v in arrays.items()}
        b = {k: v.copy() for k, v in arrays.items()}
        it = gi.run(defn, {k: (a[k], origins[k]) for k in a}, scalars, domain)
        gi.run(defn, {k: (b[k], origins[k]) for k in b}, scalars, domain, while_semantics="pointwise")
        differ = any(not np.array_equal(a[k], b[k], equal_nan=True) for k in a)
        assert not differ or it.while_reevaluates, text  # the two semantics only ever part where the counter says so
        n_while += bool(it.while_reevaluates)
        n_differ += differ
        n_quirk += bool(it.forward_section_quirk)
    print(f"of 150 random programs: {n_while} have a while loop whose body changes its own condition mid-way ({n_differ} get other values "
          f"per point than statement-wise), {n_quirk} read a temporary of an earlier merged block at a horizontal offset")
    assert n_while > 0 and n_quirk > 0  # the corpus does exercise both


# ---- hip:mi300 against the interpreter, nothing of the product on the checking side (GPU) ---------------------------------------------
def _hip_vs_interpreter(defn, externals, scalars, domain, seed, text="", **opts):
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    hip = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals or None, **opts)
    ni, nj, nk = domain
    if "region[" in text:
        ni, nj = max(ni, 6), max(nj, 6)
    domain = (ni, nj, max(nk, hip.domain_info.min_sequential_axis_size))
    arrays, origins = zoo.make_inputs(hip, domain, seed)
    expect = {k: v.copy() for k, v in arrays.items()}
    gi.run(defn, {k: (expect[k], origins[k]) for k in expect}, scalars, domain, externals=externals)
    dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k],
                                    dimensions=tuple(hip.field_info[k].axes) + tuple(str(n) for n in range(len(hip.field_info[k].data_dims))))
           for k, v in arrays.items()}
    hip(**dev, **scalars, origin=origins, domain=domain)
    for k in arrays:
        np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"field {k} (seed {seed}) {domain}\n{text}")


# GT4MI_FUZZ_SEEDS=<n> GT4MI_FUZZ_FIRST_SEED=<s>: one-off campaigns on fresh programs (results under profiles/)
_N, _FIRST = int(os.environ.get("GT4MI_FUZZ_SEEDS", "0")), int(os.environ.get("GT4MI_FUZZ_FIRST_SEED", "0"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + (_N or 100)))
def test_generated_kernels_match_the_interpreter_on_random_programs(seed, tmp_path):
    defn, scalars, text = fuzz_stencils.make_stencil(seed, tmp_path)
    for domain in (DOMAINS[seed % len(DOMAINS)], (130, 9, 6)):
        _hip_vs_interpreter(defn, {}, scalars, domain, seed, text)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(30))
def test_generated_column_kernels_match_the_interpreter_on_two_sweep_programs(seed, tmp_path):
    defn, scalars, text = fuzz_stencils.make_two_sweep_stencil(seed, tmp_path)
    _hip_vs_interpreter(defn, {}, scalars, (70, 5, 23), seed, text)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(zoo.ZOO))
def test_the_zoo_on_hip_matches_the_interpreter(name):
    defn, externals, scalars, opts = zoo.ZOO[name]
    _hip_vs_interpreter(defn, externals, scalars, (67, 9, 14), 11, **opts)


# ---- backend option while_loops="pointwise": the compiled backends' loop, checked against the interpreter's ----------------------------
def _seeds_with_a_while_loop(n):
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        return [s for s in range(n) if "while " in fuzz_stencils.make_stencil(s, pathlib.Path(d))[2]]


WHILE_SEEDS = _seeds_with_a_while_loop(150)


@pytest.mark.parametrize("seed", WHILE_SEEDS)
def test_pointwise_while_loops_match_the_interpreter(seed, tmp_path):
    """`while_loops="pointwise"`: frontend + numpy oracle, and the planner's rewritten IR under the same oracle, against the
    interpreter's per-point loop; every sixth program is also compiled for gfx950."""
    import oracle.numpy_backend as oracle_backend
    from gt4py_amd import _lib
    from gt4py_amd.cartesian import analysis, gtscript

    defn, scalars, text = fuzz_stencils.make_stencil(seed, tmp_path)
    ref = gtscript.stencil(backend="numpy", definition=defn, while_loops="pointwise")
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, while_loops="pointwise")
    ni, nj, nk = DOMAINS[seed % len(DOMAINS)]
    if "region[" in text:
        ni, nj = max(ni, 6), max(nj, 6)
    domain = (ni, nj, max(nk, ref.domain_info.min_sequential_axis_size))
    arrays, origins = zoo.make_inputs(ref, domain, seed)
    expect = {k: v.copy() for k, v in arrays.items()}
    gi.run(defn, {k: (expect[k], origins[k]) for k in expect}, scalars, domain, while_semantics="pointwise")
    got = {k: v.copy() for k, v in arrays.items()}
    ref(**got, **scalars, origin=origins, domain=domain)
    program = type(hip)._gt_program_
    rewritten = {k: v.copy() for k, v in arrays.items()}
    oracle_backend.run_stencil(program.plan.stencil, analysis.compute_extents(program.plan.stencil), domain, origins, rewritten, scalars)
    for k in arrays:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"seed {seed}, field {k}\n{text}")
        np.testing.assert_array_equal(rewritten[k], expect[k], err_msg=f"seed {seed}, field {k} (rewritten IR)\n{text}")
    if seed % 6 == 0:
        assert _lib.rtc_compile(program.source, f"pointwise_{seed}.hip", ["-DGT4MI_UNIT_I_STRIDE=1", "-DGT4MI_NO_ALIAS=1"])[:4] == b"\x7fELF"


def test_the_while_loops_option_is_validated_and_versions_the_stencil():
    from gt4py_amd.cartesian import gtscript

    defn = zoo.ZOO["newton_sqrt"][0]
    with pytest.raises(ValueError, match="while_loops"):
        gtscript.stencil(backend="hip:mi300", definition=defn, while_loops="sometimes")
    a = gtscript.stencil(backend="hip:mi300", definition=defn)
    b = gtscript.stencil(backend="hip:mi300", definition=defn, while_loops="pointwise")
    assert type(a)._gt_id_ != type(b)._gt_id_  # (the option is part of the fingerprint: two different programs)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", WHILE_SEEDS)
def test_pointwise_while_loops_on_hip_match_the_interpreter(seed, tmp_path):
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    defn, scalars, text = fuzz_stencils.make_stencil(seed, tmp_path)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, while_loops="pointwise")
    for dom in (DOMAINS[seed % len(DOMAINS)], (130, 9, 6)):
        ni, nj, nk = dom
        if "region[" in text:
            ni, nj = max(ni, 6), max(nj, 6)
        domain = (ni, nj, max(nk, hip.domain_info.min_sequential_axis_size))
        arrays, origins = zoo.make_inputs(hip, domain, seed)
        expect = {k: v.copy() for k, v in arrays.items()}
        gi.run(defn, {k: (expect[k], origins[k]) for k in expect}, scalars, domain, while_semantics="pointwise")
        dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k],
                                        dimensions=tuple(hip.field_info[k].axes) + tuple(str(n) for n in range(len(hip.field_info[k].data_dims))))
               for k, v in arrays.items()}
        hip(**dev, **scalars, origin=origins, domain=domain)
        for k in arrays:
            np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"seed {seed} {domain}, field {k}\n{text}")
