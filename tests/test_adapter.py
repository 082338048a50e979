"""gt4py OIR -> this repo's IR (gt4py_amd/adapter): the bridge for a real gt4py install.

gt4py itself cannot be imported in this image, so the OIR trees below are built by hand from plain objects that carry
the reference's node CLASS NAMES and ATTRIBUTES (gtc/oir.py:33-360, gtc/common.py:65-890) -- the only things the
translator looks at.  Each tree is the OIR the reference produces for a GTScript definition written next to it;
the translated IR must equal what this repo's own frontend makes of that definition, and must run.
"""

import numpy as np
import pytest

import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
from gt4py_amd import adapter
from gt4py_amd.cartesian import definitions as D, frontend, gtscript
from gt4py_amd.cartesian.backend import hip_backend
from gt4py_amd.cartesian.gtscript import BACKWARD, FORWARD, PARALLEL, Field, I, J, computation, horizontal, interval, region  # noqa: F401


def N(kind, **attrs):
    """An object whose class is called ``kind`` with the given attributes (an OIR-shaped node)."""
    obj = type(kind, (), {})()
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


F64, BOOL, I64 = "FLOAT64", "BOOL", "INT64"


def off(i=0, j=0, k=0):
    return N("CartesianOffset", i=i, j=j, k=k)


def field(name, i=0, j=0, k=0, dtype=F64, data_index=()):
    return N("FieldAccess", name=name, offset=off(i, j, k), dtype=dtype, data_index=list(data_index))


def lit(value, dtype=F64):
    return N("Literal", value=str(value), dtype=dtype)


def binop(op, left, right, dtype=F64):
    return N("BinaryOp", op=op, left=left, right=right, dtype=dtype)


def assign(left, right):
    return N("AssignStmt", left=left, right=right)


def bound(level, offset=0):
    return N("AxisBound", level=level, offset=offset)


def section(start, end, *hexecs):
    return N("VerticalLoopSection", interval=N("Interval", start=start, end=end), horizontal_executions=list(hexecs))


def hexec(*body, declarations=()):
    return N("HorizontalExecution", body=list(body), declarations=list(declarations))


def loop(order, *sections):
    return N("VerticalLoop", loop_order=order, sections=list(sections), caches=[])


def fdecl(name, dtype=F64, dims=(True, True, True), data_dims=()):
    return N("FieldDecl", name=name, dtype=dtype, dimensions=dims, data_dims=tuple(data_dims))


FULL = (bound("start"), bound("end"))


def parse(defn, **kw):
    return frontend.parse_stencil(defn, externals=kw.get("externals", {}), dtypes={},
                                  options=D.BuildOptions(name=defn.__name__, module=__name__, backend_opts={}))


def same_program(a, b):
    return hip_backend.canonical_form(a)[0] == hip_backend.canonical_form(b)[0]


# ---- 5-point Laplacian ------------------------------------------------------------------------------------------
def lap_definition(inp: Field[np.float64], out: Field[np.float64]):
    with computation(PARALLEL), interval(...):
        out = -4.0 * inp[0, 0, 0] + inp[-1, 0, 0] + inp[1, 0, 0] + inp[0, -1, 0] + inp[0, 1, 0]


def lap_oir():
    neg4 = N("UnaryOp", op="-", expr=lit(4.0), dtype=F64)
    rhs = binop("*", neg4, field("inp"))
    for i, j in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        rhs = binop("+", rhs, field("inp", i, j))
    return N("Stencil", name="lap", params=[fdecl("inp"), fdecl("out")], declarations=[],
             vertical_loops=[loop("parallel", section(*FULL, hexec(assign(field("out"), rhs))))])


def test_laplacian_translates_to_the_frontends_ir_and_binds_to_the_kernel_library():
    stencil, order = adapter.oir_to_ir(lap_oir())
    assert order == ("inp", "out") and same_program(stencil, parse(lap_definition))
    binding = hip_backend.recognise(stencil, D.BuildOptions(name="lap", module=__name__, backend_opts={}))
    assert binding is not None and binding.family == "lap5"


# ---- tridiagonal solve: sequential loops, several sections ---------------------------------------------------------
def tridiagonal_definition(inf: Field[np.float64], diag: Field[np.float64], sup: Field[np.float64], rhs: Field[np.float64],
                          out: Field[np.float64]):
    with computation(FORWARD):
        with interval(0, 1):
            sup = sup / diag
            rhs = rhs / diag
        with interval(1, None):
            sup = sup / (diag - sup[0, 0, -1] * inf)
            rhs = (rhs - inf * rhs[0, 0, -1]) / (diag - sup[0, 0, -1] * inf)
    with computation(BACKWARD):
        with interval(-1, None):
            out = rhs
        with interval(0, -1):
            out = rhs - sup * out[0, 0, 1]


def tridiagonal_oir():
    f = field
    den = lambda: binop("-", f("diag"), binop("*", f("sup", k=-1), f("inf")))  # noqa: E731
    forward = loop(
        "forward",
        section(bound("start"), bound("start", 1),
                hexec(assign(f("sup"), binop("/", f("sup"), f("diag")))), hexec(assign(f("rhs"), binop("/", f("rhs"), f("diag"))))),
        section(bound("start", 1), bound("end"),
                hexec(assign(f("sup"), binop("/", f("sup"), den()))),
                hexec(assign(f("rhs"), binop("/", binop("-", f("rhs"), binop("*", f("inf"), f("rhs", k=-1))), den())))))
    backward = loop(
        "backward",
        section(bound("end", -1), bound("end"), hexec(assign(f("out"), f("rhs")))),
        section(bound("start"), bound("end", -1),
                hexec(assign(f("out"), binop("-", f("rhs"), binop("*", f("sup"), f("out", k=1)))))))
    return N("Stencil", name="tridiag", params=[fdecl(n) for n in ("inf", "diag", "sup", "rhs", "out")], declarations=[],
             vertical_loops=[forward, backward])


def test_tridiagonal_translates_and_solves_on_the_oracle():
    stencil, order = adapter.oir_to_ir(tridiagonal_oir())
    assert same_program(stencil, parse(tridiagonal_definition))
    rng = np.random.default_rng(7)
    shape = (3, 4, 9)
    a = {"inf": rng.uniform(-1, 1, shape), "diag": rng.uniform(4, 5, shape), "sup": rng.uniform(-1, 1, shape),
         "rhs": rng.uniform(-10, 10, shape), "out": np.zeros(shape)}
    want = {k: v.copy() for k, v in a.items()}
    gtscript.stencil(backend="numpy", definition=tridiagonal_definition)(**want)
    adapter.stencil_class_from_ir(stencil, order, backend="numpy")()(**a)
    for k in a:
        np.testing.assert_array_equal(a[k], want[k])


# ---- masks, a while loop, a horizontal restriction, a local scalar, a cast, a native call, a scalar parameter ----------
def features_definition(a: Field[np.float64], b: Field[np.float64], *, s: float):
    with computation(PARALLEL), interval(...):
        if a > s:
            b = abs(a) + 1.0
        else:
            b = a * 2
        with horizontal(region[I[0], :]):
            b = 0.0


def features_oir():
    mask = field("mask_0", dtype=BOOL)
    body = [
        assign(field("mask_0", dtype=BOOL), binop(">", field("a"), N("ScalarAccess", name="s", dtype=F64), dtype=BOOL)),
        N("MaskStmt", mask=mask, body=[assign(field("b"), binop("+", N("NativeFuncCall", func="abs", args=[field("a")], dtype=F64), lit(1.0)))]),
        N("MaskStmt", mask=N("UnaryOp", op="not", expr=mask, dtype=BOOL),
          body=[assign(field("b"), binop("*", field("a"), N("Cast", expr=lit(2, I64), dtype=F64)))]),
    ]
    restriction = N("HorizontalRestriction",
                    mask=N("HorizontalMask", i=N("HorizontalInterval", start=bound("start", 0), end=bound("start", 1)),
                           j=N("HorizontalInterval", start=None, end=None)),
                    body=[assign(field("b"), lit(0.0))])
    return N("Stencil", name="features", params=[fdecl("a"), fdecl("b"), N("ScalarDecl", name="s", dtype=F64)],
             declarations=[N("Temporary", name="mask_0", dtype=BOOL, dimensions=(True, True, True), data_dims=())],
             vertical_loops=[loop("parallel", section(*FULL, hexec(*body), hexec(restriction)))])


def _run_features(backend, to_array):
    stencil, order = adapter.oir_to_ir(features_oir())
    rng = np.random.default_rng(3)
    a, b = rng.uniform(-2, 2, (5, 4, 3)), np.zeros((5, 4, 3))
    want_b = b.copy()
    gtscript.stencil(backend="numpy", definition=features_definition)(a.copy(), want_b, s=0.25)
    da, db = to_array(a), to_array(b)
    adapter.stencil_class_from_ir(stencil, order, backend=backend)()(da, db, s=0.25)
    return db, want_b


def test_masks_regions_and_scalars_translate_and_run_on_the_oracle():
    stencil, _ = adapter.oir_to_ir(features_oir())
    assert same_program(stencil, parse(features_definition))
    got, want = _run_features("numpy", lambda x: x.copy())
    np.testing.assert_array_equal(got, want)


def test_unknown_nodes_fail_loudly():
    bad = lap_oir()
    bad.vertical_loops[0].sections[0].horizontal_executions[0].body.append(N("SomethingNew"))
    with pytest.raises(adapter.UnsupportedOIR, match="SomethingNew"):
        adapter.oir_to_ir(bad)
    runtime = lap_oir()
    runtime.vertical_loops[0].sections[0].interval.end = N("RuntimeAxisBound", level="start", offset=None)
    with pytest.raises(adapter.UnsupportedOIR, match="run-time interval bounds"):
        adapter.oir_to_ir(runtime)


def test_registration_needs_a_real_gt4py():
    with pytest.raises(ImportError, match="needs a real gt4py install"):
        adapter.register_with_gt4py()


@pytest.mark.gpu
def test_translated_programs_run_on_hip():
    import gt4py_amd.storage as gt_storage

    def dev(x):
        return gt_storage.from_array(x, dtype=x.dtype, backend="hip:mi300", aligned_index=(0, 0, 0))

    got, want = _run_features("hip:mi300", dev)
    np.testing.assert_array_equal(gt_storage.asnumpy(got), want)
    # tridiagonal: through the kernel library (recognised from the translated IR)
    stencil, order = adapter.oir_to_ir(tridiagonal_oir())
    rng = np.random.default_rng(7)
    shape = (33, 5, 40)
    host = {"inf": rng.uniform(-1, 1, shape), "diag": rng.uniform(4, 5, shape), "sup": rng.uniform(-1, 1, shape),
            "rhs": rng.uniform(-10, 10, shape), "out": np.zeros(shape)}
    want = {k: v.copy() for k, v in host.items()}
    gtscript.stencil(backend="numpy", definition=tridiagonal_definition)(**want)
    arrays = {k: dev(v) for k, v in host.items()}
    adapter.stencil_from_oir(tridiagonal_oir())(**arrays)
    for k in host:
        np.testing.assert_array_equal(gt_storage.asnumpy(arrays[k]), want[k])
