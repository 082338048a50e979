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


import json
import pathlib
import re

#: the reference's OIR schema as data (class -> fields incl. inherited ones, enum -> members), written by scripts/make_oir_schema.py
#: from /root/reference/src/gt4py/cartesian/gtc/oir.py:28-360 and gtc/common.py:54-890 with `ast` (gt4py cannot be imported here)
SCHEMA = json.loads((pathlib.Path(__file__).parent / "golden" / "oir_schema.json").read_text())
READS = {}  # class name -> attribute names the code under test has read from nodes built here


def raw_node(kind, **attrs):
    """An object whose class is called ``kind`` with the given attributes -- NOT checked against the schema (nodes the reference
    does not have, for the tests of the translator's refusals)."""
    def reading(self, name, _kind=kind):
        if not name.startswith("__"):
            READS.setdefault(_kind, set()).add(name)
        return object.__getattribute__(self, name)

    obj = type(kind, (), {"__getattribute__": reading})()
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


def _enum_members(annotation):
    """Members (names and values) of every schema enum the annotation text mentions, or None when it mentions none."""
    allowed, found = set(), False
    if re.search(r"\bstr\b", annotation):  # (Literal.value: Union[BuiltInLiteral, str] -- any string)
        return None
    for ident in re.findall(r"[A-Za-z_]+", annotation):
        if ident in SCHEMA["enums"]:
            found = True
            for name, value in SCHEMA["enums"][ident].items():
                allowed.add(name)
                if value is not None:
                    allowed.add(str(value))
    return allowed if found else None


def N(kind, **attrs):
    """An OIR-shaped node, BUILT THROUGH THE SCHEMA: ``kind`` must be a node class of the reference's gtc/oir.py (or one of the
    gtc/common.py classes OIR uses as they are), every attribute one of that class's fields (inherited ones included), every
    required field given, and a string where the reference annotates an enum must be one of its member names or values.  The
    hand-built trees below can therefore not drift from /root/reference/src/gt4py/cartesian/gtc/oir.py:28-360."""
    assert kind in SCHEMA["classes"], f"{kind}: no such node class in the reference's OIR ({sorted(SCHEMA['classes'])})"
    fields = SCHEMA["classes"][kind]["fields"]
    unknown = set(attrs) - set(fields)
    assert not unknown, f"{kind} has no field(s) {sorted(unknown)} in the reference (it has {sorted(fields)})"
    missing = {n for n, f in fields.items() if f["required"]} - set(attrs)
    assert not missing, f"{kind}: required field(s) {sorted(missing)} not given"
    for name, value in attrs.items():
        allowed = _enum_members(fields[name]["annotation"])
        if allowed is not None and isinstance(value, str):
            assert value in allowed, f"{kind}.{name} = {value!r}: not a member of {fields[name]['annotation']}"
    return raw_node(kind, **attrs)


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
    bad.vertical_loops[0].sections[0].horizontal_executions[0].body.append(raw_node("SomethingNew"))
    with pytest.raises(adapter.UnsupportedOIR, match="SomethingNew"):
        adapter.oir_to_ir(bad)
    runtime = lap_oir()
    runtime.vertical_loops[0].sections[0].interval.end = N("RuntimeAxisBound", level="start", offset=N("ScalarAccess", name="n", dtype="INT32"))
    with pytest.raises(adapter.UnsupportedOIR, match="run-time interval bounds"):
        adapter.oir_to_ir(runtime)


# ---- a sequential column: local scalar, while, K iterator, variable and absolute K reads, a data dimension ------------------------
def column_definition(a: Field[np.float64], idx: Field[np.int64], vec: Field[(np.float64, (2,))], out: Field[np.float64]):
    with computation(FORWARD), interval(...):
        acc = a + a.at(K=0)
        while acc < 3.0:
            acc = acc + 1.0
        out = acc + K + a[0, 0, idx] + vec[0, 0, 0][1]


def column_oir():
    acc = lambda: N("ScalarAccess", name="acc", dtype=F64)  # noqa: E731
    a_at_0 = N("FieldAccess", name="a", offset=N("AbsoluteKIndex", k=0), dtype=F64, data_index=[])
    a_var = N("FieldAccess", name="a", offset=N("VariableKOffset", k=field("idx", dtype=I64)), dtype=F64, data_index=[])
    k_as_float = N("Cast", expr=N("IteratorAccess", name="K", dtype="INT32"), dtype=F64)
    body = [
        assign(acc(), binop("+", field("a"), a_at_0)),
        N("While", cond=binop("<", acc(), lit(3.0), dtype=BOOL), body=[assign(acc(), binop("+", acc(), lit(1.0)))]),
        assign(field("out"), binop("+", binop("+", binop("+", acc(), k_as_float), a_var), field("vec", data_index=[lit(1, I64)]))),
    ]
    return N("Stencil", name="column", params=[fdecl("a"), fdecl("idx", dtype=I64), fdecl("vec", data_dims=(2,)), fdecl("out")],
             declarations=[], vertical_loops=[loop("forward", section(*FULL, hexec(*body, declarations=[N("LocalScalar", name="acc", dtype=F64)])))])


def test_a_sequential_column_with_every_kind_of_k_access_translates_and_runs_on_the_oracle():
    stencil, order = adapter.oir_to_ir(column_oir())
    assert order == ("a", "idx", "vec", "out")
    rng = np.random.default_rng(11)
    shape = (4, 3, 6)
    a = rng.uniform(-1, 2, shape)
    idx = rng.integers(0, 2, shape) * (np.arange(shape[2]) < shape[2] - 1)  # k + idx stays inside the column
    vec = rng.uniform(-1, 1, shape + (2,))
    acc = a + a[:, :, :1]
    while (acc < 3.0).any():
        acc = np.where(acc < 3.0, acc + 1.0, acc)
    k = np.arange(shape[2])[None, None, :]
    want = acc + k + np.take_along_axis(a, k + idx, axis=2) + vec[..., 1]
    out = np.zeros(shape)
    adapter.stencil_class_from_ir(stencil, order, backend="numpy")()(a, idx, vec, out)
    np.testing.assert_array_equal(out, want)
    # ... and it is what this repo's own frontend makes of the GTScript spelling
    out2 = np.zeros(shape)
    gtscript.stencil(backend="numpy", definition=column_definition)(a, idx, vec, out2)
    np.testing.assert_array_equal(out2, want)


# ---- horizontal diffusion with the flux limiter, float64 and float32 fields (the reference's upcasting written out as Casts) ----------
def hdiff_oir(dt):
    """What gtir_upcaster.py:43-143 + gtir_to_oir.py make of stencil_definitions.py:316-328: literals are float64 / int64;
    a binary operation upcasts both sides to the smallest common dtype (explicit Cast nodes), a ternary its branches, an assignment its
    right-hand side to the dtype of the target; a temporary takes the dtype of its first right-hand side.  For float32 fields: the sum of
    the four neighbours and the differences in[1] - in stay float32, lap / res / flx / fly are float64, the result is rounded once."""
    F = "FLOAT64"
    wide = (lambda e: e) if dt == F else (lambda e: N("Cast", expr=e, dtype=F))
    fin = lambda i=0, j=0: field("in_field", i, j, dtype=dt)  # noqa: E731
    tmp = lambda name, i=0, j=0: field(name, i, j, dtype=F)  # noqa: E731
    nb = binop("+", binop("+", binop("+", fin(1, 0), fin(-1, 0), dt), fin(0, 1), dt), fin(0, -1), dt)
    lap = assign(tmp("lap_field"), binop("-", binop("*", lit(4.0), wide(fin())), wide(nb)))

    def flux(res_name, out_name, i, j):
        res = assign(tmp(res_name), binop("-", tmp("lap_field", i, j), tmp("lap_field")))
        cond = binop(">", binop("*", tmp(res_name), wide(binop("-", fin(i, j), fin(), dt))), N("Cast", expr=lit(0, I64), dtype=F), dtype=BOOL)
        return res, assign(tmp(out_name), N("TernaryOp", cond=cond, true_expr=N("Cast", expr=lit(0, I64), dtype=F), false_expr=tmp(res_name), dtype=F))

    res_x, flx = flux("res", "flx_field", 1, 0)
    res_y, fly = flux("res", "fly_field", 0, 1)  # (the same temporary `res`, assigned a second time: as written in the definition)
    div = binop("-", binop("+", binop("-", tmp("flx_field"), tmp("flx_field", -1, 0)), tmp("fly_field")), tmp("fly_field", 0, -1))
    rhs = binop("-", wide(fin()), binop("*", wide(field("coeff", dtype=dt)), div))
    out = assign(field("out_field", dtype=dt), rhs if dt == F else N("Cast", expr=rhs, dtype=dt))
    temps = [N("Temporary", name=n, dtype=F, dimensions=(True, True, True), data_dims=()) for n in ("lap_field", "res", "flx_field", "fly_field")]
    return N("Stencil", name="hdiff", params=[fdecl("in_field", dt), fdecl("out_field", dt), fdecl("coeff", dt)], declarations=temps,
             vertical_loops=[loop("parallel", section(*FULL, *(hexec(st) for st in (lap, res_x, flx, res_y, fly, out))))])


@pytest.mark.parametrize("dt,np_dtype", [("FLOAT64", np.float64), ("FLOAT32", np.float32)])
def test_horizontal_diffusion_translates_and_matches_the_frontend_bit_for_bit(dt, np_dtype):
    """The third stencil of the hot path, through a tree that this repository's FRONTEND never saw: the translated program must give
    exactly what the frontend's parse of the GTScript definition gives (the float32 case exercises every upcasting rule of SURVEY N2)
    -- a check of parser, dtype resolution and extent analysis that does not share them."""
    from gt4py_amd.cartesian.backend import hip_templates

    stencil, order = adapter.oir_to_ir(hdiff_oir(dt))
    assert order == ("in_field", "out_field", "coeff")
    rng = np.random.default_rng(11)
    shape, dom = (21, 18, 3), (17, 14, 3)
    a = (5.0 + rng.uniform(-1, 1, shape)).astype(np_dtype)
    c = rng.uniform(0, 0.05, shape).astype(np_dtype)
    want, got = np.zeros(shape, np_dtype), np.zeros(shape, np_dtype)
    gtscript.stencil(backend="numpy", definition=hip_templates.hdiff_limiter_field, dtypes={"T": np_dtype})(a, want, c, origin=(2, 2, 0), domain=dom)
    adapter.stencil_class_from_ir(stencil, order, backend="numpy")()(a, got, c, origin=(2, 2, 0), domain=dom)
    np.testing.assert_array_equal(got, want)
    assert np.count_nonzero(want) == np.prod(dom)  # (the limiter fires on part of the points and the interior is written everywhere)
    # the extents the analysis derives from the translated program are the reference's: in_field read 2 deep, everything else 0
    info = adapter.stencil_class_from_ir(stencil, order, backend="numpy")().field_info
    assert info["in_field"].boundary[:2] == ((2, 2), (2, 2)) and info["coeff"].boundary[:2] == ((0, 0), (0, 0))
    # ... and on hip:mi300 the translated program binds to the hand-written kernel family, like the frontend's
    binding = hip_backend.recognise(stencil, D.BuildOptions(name="hdiff", module=__name__, backend_opts={}))
    assert binding is not None and binding.family == "hdiff"


# ---- the bridge is pinned to the reference's schema -------------------------------------------------------------------------
def test_the_hand_built_trees_cannot_drift_from_the_references_oir_schema():
    """`N(...)` refuses what gtc/oir.py does not have ..."""
    with pytest.raises(AssertionError, match="no such node class"):
        N("HorizontalLoop", body=[])
    with pytest.raises(AssertionError, match=r"has no field\(s\) \['stages'\]"):
        N("VerticalLoopSection", interval=N("Interval", start=bound("start"), end=bound("end")), stages=[])
    with pytest.raises(AssertionError, match="required field"):
        N("FieldAccess", name="a", dtype=F64)  # no offset
    with pytest.raises(AssertionError, match="not a member"):
        N("VerticalLoop", loop_order="sideways", sections=[])
    with pytest.raises(AssertionError, match="not a member"):
        binop("+=", field("a"), field("b"))
    with pytest.raises(AssertionError, match="not a member"):
        fdecl("a", dtype="FLOAT16")
    # ... the schema is the one scripts/make_oir_schema.py writes from the reference's two files (checked where they exist:
    # /root/reference does not travel to the GPU box)
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parent.parent
    if pathlib.Path("/root/reference/src/gt4py/cartesian/gtc/oir.py").exists():
        proc = subprocess.run([sys.executable, str(root / "scripts" / "make_oir_schema.py"), "--check"], capture_output=True, text=True)
        assert proc.returncode == 0, proc.stderr
    assert {"oir.py", "common.py"} == set(SCHEMA["sources"]) and len(SCHEMA["classes"]) >= 35


def test_the_translator_reads_only_fields_the_reference_has():
    """... and `oir_to_ir` reads nothing else: (1) every attribute it read from a node built here, in all of this file's trees, is a
    field of that node's class in the reference; (2) every attribute name that appears in the translator's source at all is a
    field of SOME OIR class or a member of the Python enum protocol (`name` / `value`)."""
    import ast
    import inspect

    READS.clear()
    for tree in (lap_oir(), tridiagonal_oir(), features_oir(), column_oir()):
        adapter.oir_to_ir(tree)
    assert {"Stencil", "VerticalLoop", "VerticalLoopSection", "HorizontalExecution", "Interval", "AxisBound", "FieldAccess", "CartesianOffset",
            "VariableKOffset", "AbsoluteKIndex", "IteratorAccess", "LocalScalar", "While", "MaskStmt", "HorizontalRestriction",
            "HorizontalMask", "HorizontalInterval", "Cast", "NativeFuncCall", "Temporary", "ScalarDecl", "FieldDecl"} <= set(READS)
    for kind, names in READS.items():
        # (`name` / `value` probes of `_enum_name` / `_enum_value` on things that may be enums are the enum protocol, not OIR fields)
        extra = names - set(SCHEMA["classes"][kind]["fields"]) - {"name", "value"}
        assert not extra, f"oir_to_ir read {sorted(extra)} from a {kind}: the reference's class has {sorted(SCHEMA['classes'][kind]['fields'])}"
    every_field = {f for c in SCHEMA["classes"].values() for f in c["fields"]} | {"name", "value"}
    source = ast.parse(inspect.getsource(adapter._Translator))
    node_vars = {"e", "off", "stmt", "decl", "loop", "section", "hexec", "local", "st", "b", "iv", "left", "d", "t", "level"}
    read = {n.attr for n in ast.walk(source) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id in node_vars}
    read |= {n.attr for n in ast.walk(source) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Attribute)
             and isinstance(n.value.value, ast.Name) and n.value.value.id in node_vars}  # stmt.mask.i, section.interval.start
    assert read and not (read - every_field), f"attribute(s) {sorted(read - every_field)} are fields of no class of the reference's OIR"


def test_the_gt4py_facing_glue_uses_only_names_the_reference_has():
    """`register_with_gt4py` / `_wrap_for_gt4py` cannot run here (no gt4py: Python 3.10).  What can be checked: every attribute
    chain in their source that starts at a gt4py object -- the builder, its options / stencil id / backend, the imported modules,
    the args data -- names something the reference defines, `_call_run` is called with keywords it takes, and the generated
    `StencilObject` subclass defines members `StencilObject` declares.  The surface is data written from the reference's sources
    by scripts/make_gt4py_api_surface.py."""
    import ast
    import inspect
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parent.parent
    api = json.loads((root / "tests" / "golden" / "gt4py_api_surface.json").read_text())["modules"]
    if pathlib.Path("/root/reference/src/gt4py/cartesian/backend/base.py").exists():
        proc = subprocess.run([sys.executable, str(root / "scripts" / "make_gt4py_api_surface.py"), "--check"], capture_output=True, text=True)
        assert proc.returncode == 0, proc.stderr

    def members(module, cls):
        c = api[module]["classes"][cls]
        own = set(c["attributes"]) | set(c["properties"]) | set(c["methods"])
        for base in c["bases"]:
            base = base.split("[")[0].split(".")[-1]
            for mod in api.values():
                if base in mod["classes"]:
                    own |= members(next(m for m in api if api[m] is mod), base)
        return own

    def module_names(module):
        m = api[module]
        return set(m["functions"]) | set(m["names"]) | set(m["imports"]) | set(m["classes"])

    builder = members("stencil_builder.py", "StencilBuilder")
    chains_allowed = {
        ("builder",): builder, ("self", "builder"): builder,
        ("builder", "options"): members("definitions.py", "BuildOptions"), ("self", "builder", "options"): members("definitions.py", "BuildOptions"),
        ("builder", "stencil_id"): members("definitions.py", "StencilID"),
        ("builder", "backend"): members("backend/base.py", "Backend"),
        ("gt4py_base",): module_names("backend/base.py"), ("gt4py_backend",): module_names("backend/__init__.py"),
        ("gtc_passes",): module_names("gtc/passes/__init__.py"),
        ("args_data",): members("backend/module_generator.py", "ModuleData"),
    }
    assert "as_dict" in chains_allowed[("builder", "options")] and "backend_name" not in builder  # (the bug this check found)

    def chain(node):
        out = []
        while isinstance(node, ast.Attribute):
            out.append(node.attr)
            node = node.value
        return (node.id,) + tuple(reversed(out)) if isinstance(node, ast.Name) else None

    seen = set()
    for fn in (adapter.register_with_gt4py, adapter._wrap_for_gt4py):
        tree = ast.parse(inspect.getsource(fn))
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute):
                c = chain(node)
                if c is None:
                    continue
                for n in range(len(c) - 1, 0, -1):  # the longest known prefix decides
                    if c[:n] in chains_allowed:
                        assert c[n] in chains_allowed[c[:n]], f"{'.'.join(c[:n + 1])}: the reference has no such name (it has {sorted(chains_allowed[c[:n]])[:40]})"
                        seen.add(c[:n + 1])
                        break
            if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("gt4py.cartesian"):
                rel = node.module[len("gt4py.cartesian."):].replace(".", "/") if node.module != "gt4py.cartesian" else ""
                for alias in node.names:
                    if rel == "":  # from gt4py.cartesian import backend
                        assert (alias.name + "/__init__.py") in api or (alias.name + ".py") in api, alias.name
                    elif (rel + "/" + alias.name + ".py") in api or (rel + "/" + alias.name + "/__init__.py") in api:
                        pass  # a submodule: from gt4py.cartesian.backend import base
                    else:
                        module = rel + ".py" if (rel + ".py") in api else rel + "/__init__.py"
                        assert alias.name in module_names(module), f"from {node.module} import {alias.name}: not defined there"
                        seen.add((node.module, alias.name))
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "_call_run":
                p = api["stencil_object.py"]["classes"]["StencilObject"]["methods"]["_call_run"]
                assert {k.arg for k in node.keywords} <= set(p["positional"]) | set(p["keyword_only"])
                assert set(p["positional"]) - {"self"} <= {k.arg for k in node.keywords}
                seen.add(("_call_run",))
    assert {("builder", "gtir_pipeline"), ("builder", "backend", "name"), ("builder", "options", "as_dict"), ("builder", "stencil_id", "version"),
            ("self", "builder", "gtir"), ("gt4py_base", "BaseBackend"), ("gt4py_backend", "register"), ("gtc_passes", "OirPipeline"),
            ("args_data", "field_info"), ("_call_run",), ("gt4py.cartesian.backend.module_generator", "make_args_data_from_gtir"),
            ("gt4py.cartesian.gtc.gtir_to_oir", "GTIRToOIR"), ("gt4py.cartesian.stencil_object", "StencilObject")} <= seen, sorted(seen)
    # what the Backend subclass sets and overrides, and what the StencilObject subclass defines, are members the base classes declare
    source = inspect.getsource(adapter.register_with_gt4py)
    backend_members = members("backend/base.py", "BaseBackend")
    for name in ("options", "storage_info", "languages", "name", "generate", "load", "check_options", "builder"):
        assert name in backend_members and name in source
    stencil_members = members("stencil_object.py", "StencilObject")
    wrap = ast.parse(inspect.getsource(adapter._wrap_for_gt4py))
    keys = [k.value for n in ast.walk(wrap) if isinstance(n, ast.Dict) for k in n.keys if isinstance(k, ast.Constant)]
    assert set(keys) - {"__module__"} <= stencil_members | {"__call__"}, sorted(set(keys) - stencil_members)
    assert {"run", "backend", "source", "domain_info", "field_info", "parameter_info", "constants", "options", "_gt_id_", "definition_func"} <= set(keys)
    # make_args_data_from_gtir takes the builder's GTIR pipeline (module_generator.py:56)
    assert api["backend/module_generator.py"]["functions"]["make_args_data_from_gtir"]["positional"] == ["pipeline"]


def _run_the_glue_on_the_double(monkeypatch):
    """register_with_gt4py + generate() + the StencilObject subclass of _wrap_for_gt4py, against tests/gt4py_double.py."""
    import gt4py_double
    from gt4py_amd.cartesian import analysis

    root = pathlib.Path(__file__).resolve().parent.parent
    surface = json.loads((root / "tests" / "golden" / "gt4py_api_surface.json").read_text())["modules"]
    mods = gt4py_double.build(surface, lambda pipeline: analysis.make_args_data(adapter.oir_to_ir(pipeline.oir)[0]))
    gt4py_double.check_against_surface(mods, surface)
    gt4py_double.install(monkeypatch, mods)
    backend_cls = adapter.register_with_gt4py("hip:mi300")
    assert mods["gt4py.cartesian.backend.base"].REGISTRY["hip:mi300"] is backend_cls and backend_cls.name == "hip:mi300"
    assert tuple(backend_cls.storage_info["layout_map"](("I", "J", "K"))) == (2, 1, 0) and backend_cls.storage_info["alignment"] == 32
    builder = gt4py_double.StencilBuilder(tridiagonal_definition, tridiagonal_oir(), name="tridiagonal_definition", backend_opts={"device_sync": True})
    builder.backend = backend_cls(builder)
    stencil_class = builder.backend.generate()
    assert builder.backend.load() is None
    return mods, builder, stencil_class


def test_the_gt4py_facing_glue_executes_against_a_double_of_the_references_interface(monkeypatch):
    """The ~40 lines that subclass gt4py's BaseBackend and StencilObject RUN here, on a double that offers exactly the names the pinned
    API surface lists (tests/gt4py_double.py): registration, option checking, OIR -> IR -> a stencil bound to the kernel library,
    the generated StencilObject subclass with its properties.  (Calling it needs the device: the GPU twin below.)"""
    mods, builder, stencil_class = _run_the_glue_on_the_double(monkeypatch)
    StencilObject = mods["gt4py.cartesian.stencil_object"].StencilObject
    assert issubclass(stencil_class, StencilObject) and stencil_class.__name__ == builder.class_name and stencil_class.__module__ == builder.module_qualname
    obj = stencil_class()
    assert obj is stencil_class()  # gt4py's singleton protocol, inherited
    assert obj.backend == "hip:mi300" and obj._gt_id_ == builder.stencil_id.version and obj.definition_func is tridiagonal_definition
    assert set(obj.field_info) == {"inf", "diag", "sup", "rhs", "out"} and obj.parameter_info == {} and obj.domain_info.min_sequential_axis_size == 2
    assert obj.field_info["sup"].access.name == "READ_WRITE" and obj.options["name"] == "tridiagonal_definition" and obj.constants == {}
    assert "sup" in obj.source and "rhs" in obj.source
    # an option the backend does not declare is reported the way gt4py's BaseBackend.check_options reports it
    bad = __import__("gt4py_double").StencilBuilder(lap_definition, lap_oir(), name="lap", backend_opts={"no_such_option": 1})
    bad.backend = mods["gt4py.cartesian.backend.base"].REGISTRY["hip:mi300"](bad)
    with pytest.warns(RuntimeWarning, match="no_such_option"):
        bad.backend.generate()


@pytest.mark.gpu
def test_the_glue_on_the_double_computes_on_the_device(monkeypatch):
    """... and a call through the generated class -- gt4py's `_call_run` protocol (here the double's) -> `run` -> this repository's
    implementation -> the C ABI -> the tridiagonal kernel -- gives the oracle's values."""
    import gt4py_amd.storage as gt_storage

    _, _, stencil_class = _run_the_glue_on_the_double(monkeypatch)
    rng = np.random.default_rng(7)
    shape = (33, 5, 40)
    host = {"inf": rng.uniform(-1, 1, shape), "diag": rng.uniform(4, 5, shape), "sup": rng.uniform(-1, 1, shape),
            "rhs": rng.uniform(-10, 10, shape), "out": np.zeros(shape)}
    want = {k: v.copy() for k, v in host.items()}
    gtscript.stencil(backend="numpy", definition=tridiagonal_definition)(**want)
    arrays = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=(0, 0, 0)) for k, v in host.items()}
    # what the real `_call_run` hands to `run()` is NOT the caller's object: every field went through `cp.asarray` first
    # (/root/reference/src/gt4py/cartesian/stencil_object.py:69-93, 585-609 -> storage/cartesian/utils.py:176-215).  The double does the
    # same with a stand-in that exposes shape / dtype / strides / __cuda_array_interface__ and nothing else.
    seen = {}
    inner_run = stencil_class.run

    def recording_run(self, _domain_, _origin_, exec_info=None, **kwargs):
        seen.update({k: type(v).__name__ for k, v in kwargs.items()})
        return inner_run(self, _domain_, _origin_, exec_info=exec_info, **kwargs)

    monkeypatch.setattr(stencil_class, "run", recording_run)
    stencil_class()(arrays["inf"], arrays["diag"], arrays["sup"], rhs=arrays["rhs"], out=arrays["out"])  # positional and keyword fields
    assert seen == {k: "CupyLikeArray" for k in host}
    for k in host:
        np.testing.assert_array_equal(gt_storage.asnumpy(arrays[k]), want[k])
    # a strided view (cupy keeps the strides of what it wraps): the K-reversed... no: a J-sliced view with an origin offset
    view_host = {k: v.copy() for k, v in host.items()}
    want2 = {k: v[:, 1:4, :].copy() for k, v in view_host.items()}
    gtscript.stencil(backend="numpy", definition=tridiagonal_definition)(**want2)
    arrays2 = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=(0, 0, 0)) for k, v in view_host.items()}
    views = {k: a[:, 1:4, :] for k, a in arrays2.items()}  # non-contiguous device views: strides must survive the conversion
    stencil_class()(**views)
    for k in host:
        got = gt_storage.asnumpy(arrays2[k])
        np.testing.assert_array_equal(got[:, 1:4, :], want2[k])
        np.testing.assert_array_equal(got[:, :1, :], view_host[k][:, :1, :])  # rows outside the view untouched
        np.testing.assert_array_equal(got[:, 4:, :], view_host[k][:, 4:, :])


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
