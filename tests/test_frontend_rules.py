"""GTScript front-end rules the `hip:mi300` path depends on, one builder-authored snippet per rule.

Every case names the rule of the reference's parser it pins (file:line in
/root/reference/src/gt4py/cartesian/frontend/gtscript_frontend.py unless another file is given); the snippets, field names and
expected values are this repo's own.  The numerical behaviour of what the front end accepts is pinned elsewhere (the reference's
suites in tests/reference_suites.py, the fuzzers against the independent interpreter); here: what parses, what is refused, and
what the parse tree records.
"""

import enum
import types

import numpy as np
import pytest

from gt4py_amd.cartesian import definitions as D, frontend, gtscript, ir
from gt4py_amd.cartesian.definitions import GTScriptDefinitionError, GTScriptSymbolError, GTScriptSyntaxError
from gt4py_amd.cartesian.gtscript import (  # noqa: F401
    BACKWARD, FORWARD, IJ, IJK, PARALLEL, Field, I, J, K, compile_assert, computation, exp, horizontal, interval, log, region, sqrt,
)


def parse(definition, *, externals=None, dtypes=None, **options):
    opts = D.BuildOptions(name=definition.__name__, module=__name__, backend_opts={}, **options)
    return frontend.parse_stencil(definition, externals=externals or {}, dtypes=dtypes or {}, options=opts)


def assignments(stencil):
    return [s for _, _, s in stencil.statements()]


def literal_values(node):
    return [e.value for e in ir.walk(node) if isinstance(e, ir.Literal)]


def refused(definition, error=GTScriptSyntaxError, match=None, **kw):
    with pytest.raises(error, match=match):
        parse(definition, **kw)


GRAVITY = 9.80665
PHYS = types.SimpleNamespace(cp=1004.5, gas=types.SimpleNamespace(rd=287.05))
LEVELS_I32 = np.int32(60)
WEIGHT_F32 = np.float32(0.25)


# ---- module-level constants are inlined as literals (ValueInliner :416-470; resolution of the names GTScriptParser :2311-2380) ----
def test_module_constants_and_nested_namespaces_become_literals():
    def buoyancy(theta: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b = GRAVITY * theta / PHYS.cp + PHYS.gas.rd

    (stmt,) = assignments(parse(buoyancy))
    assert set(literal_values(stmt.value)) == {GRAVITY, 1004.5, 287.05}


def test_numpy_typed_constants_keep_their_own_precision():
    # a numpy scalar is typed by its dtype, not by literal_*_precision (visit_Constant :1236-1266)
    def scale(q: Field[np.float32]):
        with computation(PARALLEL), interval(...):
            n = LEVELS_I32
            w = WEIGHT_F32
            q = q * w + n

    n_stmt, w_stmt, _ = assignments(parse(scale, literal_float_precision=64, literal_int_precision=64))
    assert n_stmt.value.dtype == np.dtype(np.int32) and w_stmt.value.dtype == np.dtype(np.float32)


def test_an_unknown_name_is_a_symbol_error_and_a_missing_attribute_a_definition_error():
    def uses_nothing(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = a + NOT_DEFINED_ANYWHERE  # noqa: F821

    refused(uses_nothing, GTScriptSymbolError, "NOT_DEFINED_ANYWHERE")

    def uses_missing_attribute(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = a + PHYS.latent_heat

    refused(uses_missing_attribute, GTScriptDefinitionError, "PHYS.latent_heat")


@pytest.mark.parametrize("bad", ["a string", {"k": 1}, [1.0, 2.0]])
def test_a_constant_that_is_not_a_number_is_refused(bad):
    TABLE = bad

    def uses_table(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = a + TABLE

    refused(uses_table, GTScriptDefinitionError, "TABLE")


# ---- externals (GTScriptParser.resolve_external_symbols :2311-2380; `from __externals__ import` :2283-2308) -----------------------
def test_externals_are_imported_inside_the_definition_and_a_missing_one_is_refused():
    def relax(t: Field[float], t_eq: Field[float]):
        from __externals__ import TAU

        with computation(PARALLEL), interval(...):
            t = t - (t - t_eq) / TAU

    (stmt,) = assignments(parse(relax, externals={"TAU": 3600.0}))
    assert 3600.0 in literal_values(stmt.value)
    refused(relax, GTScriptDefinitionError, "TAU")  # no externals given


def test_only_externals_may_be_imported():
    def imports_os(a: Field[float]):
        import os  # noqa: F401

        with computation(PARALLEL), interval(...):
            a = 0.0

    refused(imports_os, GTScriptSyntaxError, "import")


def test_a_boolean_external_selects_a_branch_at_parse_time():
    # CompiledIfInliner :765-800: an `if` on an external is resolved while parsing; the other branch is never looked at
    def saturate(q: Field[float], qs: Field[float]):
        from __externals__ import CLIP

        with computation(PARALLEL), interval(...):
            if __INLINED(CLIP):  # noqa: F821
                q = qs if q > qs else q
            else:
                q = q + UNDEFINED_IN_DEAD_BRANCH  # noqa: F821

    stmts = assignments(parse(saturate, externals={"CLIP": np.bool_(True)}))
    assert len(stmts) == 1 and isinstance(stmts[0].value, ir.TernaryOp)


# ---- gtscript.function (CallInliner :505-760; ReturnReplacer :472-500) ---------------------------------------------------------------
@gtscript.function
def centred_x(f):
    return 0.5 * (f[1, 0, 0] - f[-1, 0, 0])


@gtscript.function
def gradient(f):
    gx = centred_x(f)
    gy = 0.5 * (f[0, 1, 0] - f[0, -1, 0])
    return gx, gy


def test_functions_are_inlined_offsets_compose_and_tuples_unpack():
    def slope(h: Field[float], sx: Field[float], sy: Field[float]):
        with computation(PARALLEL), interval(...):
            sx, sy = gradient(h[1, 0, 0])

    st = parse(slope)
    stmts = assignments(st)
    assert [s.target.name for s in stmts][-2:] == ["sx", "sy"]
    # the callees are gone: every statement is an assignment of the caller, and the argument's own offset (+1 in I) is added to the
    # offsets inside the callees (directly or through the temporaries the inliner makes for the arguments)
    def reach(name, off, seen=()):
        out = set()
        for s in stmts:
            if s.target.name == name:
                for r in ir.walk(s.value):
                    if isinstance(r, ir.FieldAccess):
                        total = tuple(a + b for a, b in zip(off, r.offset))
                        out |= {total} if r.name == "h" else reach(r.name, total, seen + (name,)) if r.name not in seen else set()
        return out

    assert reach("sx", (0, 0, 0)) == {(2, 0, 0), (0, 0, 0)} and reach("sy", (0, 0, 0)) == {(1, 1, 0), (1, -1, 0)}


def test_a_plain_python_function_cannot_be_called():
    def helper(x):
        return x

    def calls_helper(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = helper(a)

    refused(calls_helper, (GTScriptSyntaxError, TypeError))


def test_function_argument_rules():
    @gtscript.function
    def blend(a, b, *, w=0.5):
        return w * a + (1.0 - w) * b

    def ok(x: Field[float], y: Field[float], z: Field[float]):
        with computation(PARALLEL), interval(...):
            z = blend(x, y) + blend(x, y, w=0.25)

    (stmt,) = [s for s in assignments(parse(ok)) if s.target.name == "z"]
    assert stmt is not None

    def missing_argument(x: Field[float], z: Field[float]):
        with computation(PARALLEL), interval(...):
            z = blend(x)

    refused(missing_argument, GTScriptSyntaxError)

    def unknown_keyword(x: Field[float], y: Field[float], z: Field[float]):
        with computation(PARALLEL), interval(...):
            z = blend(x, y, weight=0.1)

    refused(unknown_keyword, GTScriptSyntaxError)


def test_the_number_of_returned_values_must_match_the_targets():
    def two_from_one(h: Field[float], sx: Field[float], sy: Field[float]):
        with computation(PARALLEL), interval(...):
            sx, sy = centred_x(h)

    refused(two_from_one, GTScriptSyntaxError, "return")


def test_a_function_sees_the_globals_of_the_moment_it_was_decorated():
    FACTOR = 2.0

    @gtscript.function
    def doubled(x):
        return FACTOR * x

    FACTOR = 3.0  # noqa: F841 -- too late: `doubled` captured 2.0

    def use(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = doubled(a)

    values = [v for s in assignments(parse(use)) for v in literal_values(s.value)]
    assert 2.0 in values and 3.0 not in values


# ---- offsets (visit_Subscript :1407-1475; axis syntax gtscript.AxisIndex) -----------------------------------------------------------------
def test_axis_offsets_are_the_same_as_index_tuples():
    def shifted(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b = a[I + 1] + a[J - 2] + a[K + 1] + a[I - 1, K + 1]

    (stmt,) = assignments(parse(shifted))
    offs = sorted(tuple(r.offset) for r in ir.walk(stmt.value) if isinstance(r, ir.FieldAccess))
    assert offs == sorted([(1, 0, 0), (0, -2, 0), (0, 0, 1), (-1, 0, 1)])


def test_offset_errors():
    def repeated_axis(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b = a[I + 1, I - 1]

    refused(repeated_axis, GTScriptSyntaxError)

    def too_few_indices(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b = a[1, 0]

    refused(too_few_indices, GTScriptSyntaxError)

    def non_constant_horizontal(a: Field[float], b: Field[float], n: int):
        with computation(PARALLEL), interval(...):
            b = a[n, 0, 0]

    refused(non_constant_horizontal, GTScriptSyntaxError)


def test_a_written_target_carries_no_offset():
    def offset_target(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b[1, 0, 0] = a

    refused(offset_target, GTScriptSyntaxError)

    def zero_offset_target(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b[0, 0, 0] = a

    assert len(assignments(parse(zero_offset_target))) == 1  # the all-zero subscript is the field itself (visit_Assign :1822-1850)


# ---- computation / interval structure (visit_With :1946-2030; IntervalParser :105-227) ------------------------------------------------------
def test_intervals_of_one_computation_are_kept_in_iteration_order():
    def column(a: Field[float]):
        with computation(BACKWARD):
            with interval(-1, None):
                a = 0.0
            with interval(0, -1):
                a = a[0, 0, 1] + 1.0

    st = parse(column)
    (comp,) = st.computations
    assert comp.order == ir.LoopOrder.BACKWARD
    assert [(b.interval.start.level, b.interval.start.offset) for b in comp.blocks] == [(ir.Level.END, -1), (ir.Level.START, 0)]


def test_interval_syntax_errors():
    def overlapping(a: Field[float]):
        with computation(FORWARD):
            with interval(0, 3):
                a = 1.0
            with interval(2, None):
                a = 2.0

    refused(overlapping, GTScriptSyntaxError, "[Oo]verlap")

    def wrong_order_for_forward(a: Field[float]):
        with computation(FORWARD):
            with interval(1, None):
                a = 1.0
            with interval(0, 1):
                a = 2.0

    refused(wrong_order_for_forward, GTScriptSyntaxError)

    def three_bounds(a: Field[float]):
        with computation(PARALLEL), interval(0, 1, 2):
            a = 1.0

    refused(three_bounds, GTScriptSyntaxError)

    def no_interval(a: Field[float]):
        with computation(PARALLEL):
            a = 1.0

    refused(no_interval, GTScriptSyntaxError)

    def statement_outside(a: Field[float]):
        a = 1.0
        with computation(PARALLEL), interval(...):
            a = 2.0

    refused(statement_outside, GTScriptSyntaxError)


def test_interval_bounds_may_be_externals():
    def capped(a: Field[float]):
        from __externals__ import KTOP

        with computation(PARALLEL), interval(KTOP, None):
            a = 1.0

    (comp,) = parse(capped, externals={"KTOP": 3}).computations
    assert (comp.blocks[0].interval.start.level, comp.blocks[0].interval.start.offset) == (ir.Level.START, 3)
    refused(capped, GTScriptSyntaxError, externals={"KTOP": 2.5})  # a level is an integer (IntervalParser.visit_Constant :162-173)


def test_nested_with_forms_are_equivalent():
    def flat(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = 1.0

    def nested(a: Field[float]):
        with computation(PARALLEL):
            with interval(...):
                a = 1.0

    f, n = parse(flat), parse(nested)
    assert len(f.computations) == len(n.computations) == 1 and len(assignments(f)) == len(assignments(n)) == 1

    def interval_outside_computation(a: Field[float]):
        with interval(...):
            with computation(PARALLEL):
                a = 1.0

    refused(interval_outside_computation, GTScriptSyntaxError)


# ---- horizontal regions (HorizontalIntervalParser :229-300; visit_With :1975-2003) ----------------------------------------------------------
def test_regions_record_their_bounds_relative_to_the_domain_edges():
    def edges(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = 0.0
            with horizontal(region[I[0], :], region[I[-1] - 1, J[0] : J[-1]]):
                a = 1.0

    whole, west, east = assignments(parse(edges))  # one statement per region of a `horizontal` block
    assert whole.region is None
    assert (west.region.i.start, west.region.i.end) == (ir.AxisBound(ir.Level.START, 0), ir.AxisBound(ir.Level.START, 1))
    assert west.region.j.start is None and west.region.j.end is None  # `:` = the whole axis
    assert (east.region.i.start, east.region.i.end) == (ir.AxisBound(ir.Level.END, -2), ir.AxisBound(ir.Level.END, -1))
    assert (east.region.j.start, east.region.j.end) == (ir.AxisBound(ir.Level.START, 0), ir.AxisBound(ir.Level.END, -1))


def test_region_errors():
    def region_of_k(a: Field[float]):
        with computation(PARALLEL), interval(...):
            with horizontal(region[K[0], :]):
                a = 1.0

    refused(region_of_k, GTScriptSyntaxError)

    def region_in_sequential_offset_read(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            with horizontal(region[:, :, :]):
                a = b

    refused(region_in_sequential_offset_read, GTScriptSyntaxError)


# ---- assignments (visit_Assign :1789-1935) -----------------------------------------------------------------------------------------------
def test_what_may_stand_on_the_left():
    def to_parameter(a: Field[float], dt: float):
        with computation(PARALLEL), interval(...):
            dt = 1.0
            a = dt

    refused(to_parameter, GTScriptSyntaxError)

    def chained(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            a = b = 1.0

    refused(chained, GTScriptSyntaxError)

    def augmented(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            a += b * 2.0

    (stmt,) = assignments(parse(augmented))  # `a += x` is `a = a + x` (visit_AugAssign :1938-1944)
    assert isinstance(stmt.value, ir.BinaryOp) and stmt.value.op == "+"


def test_a_temporary_must_be_written_before_it_is_read():
    def reads_first(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = tmp + 1.0  # noqa: F821
            tmp = a  # noqa: F841

    refused(reads_first, (GTScriptSymbolError, GTScriptSyntaxError))


def test_temporaries_take_the_dtype_of_their_first_value_or_of_their_annotation():
    def typed(a: Field[np.float32], out: Field[np.float64]):
        with computation(PARALLEL), interval(...):
            narrow = a * a            # float32 * float32
            wide: np.float64 = a      # annotated
            out = narrow + wide

    st = parse(typed)
    temps = {t.name: np.dtype(t.dtype) for t in st.temporaries}
    assert temps == {"narrow": np.dtype(np.float32), "wide": np.dtype(np.float64)}


# ---- control flow (visit_If :1604-1655, visit_While :1656-1730) ---------------------------------------------------------------------------
def test_if_and_while_conditions():
    def clipped(a: Field[float], lo: float, hi: float):
        with computation(PARALLEL), interval(...):
            if a < lo:
                a = lo
            elif a > hi:
                a = hi

    assert len(assignments(parse(clipped))) >= 2

    def newton(x: Field[float], y: Field[float]):
        with computation(PARALLEL), interval(...):
            r = y
            n = 0
            while n < 4:
                r = 0.5 * (r + x / r)
                n = n + 1
            y = r

    looped = [s for s in assignments(parse(newton)) if s.loops]
    assert {s.target.name for s in looped} == {"r", "n"} and len({s.loops[0][0] for s in looped}) == 1  # one loop, two body statements

    def while_else(x: Field[float]):
        with computation(PARALLEL), interval(...):
            while x > 1.0:
                x = x * 0.5
            else:
                x = 0.0

    refused(while_else, GTScriptSyntaxError)


def test_unsupported_python_statements_are_refused():
    def has_for(a: Field[float]):
        with computation(PARALLEL), interval(...):
            for n in range(3):  # noqa: B007
                a = a + 1.0

    refused(has_for, GTScriptSyntaxError)

    def has_return(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = 1.0
            return a

    refused(has_return, GTScriptSyntaxError)


# ---- compile_assert (AssertionChecker :60-102) ----------------------------------------------------------------------------------------------
def test_compile_assert_is_checked_while_parsing():
    def guarded(a: Field[float]):
        from __externals__ import ORDER

        with computation(PARALLEL), interval(...):
            compile_assert(ORDER >= 2)
            a = a * ORDER

    assert assignments(parse(guarded, externals={"ORDER": 4}))
    refused(guarded, frontend.GTScriptAssertionError if hasattr(frontend, "GTScriptAssertionError") else Exception, externals={"ORDER": 1})

    def on_a_field(a: Field[float]):
        with computation(PARALLEL), interval(...):
            compile_assert(a > 0.0)
            a = 1.0

    refused(on_a_field, GTScriptSyntaxError)


# ---- signatures and dtypes (GTScriptParser.annotate_definition :2136-2260; field descriptors gtscript.py) ---------------------------------------
def test_signature_rules():
    def unannotated(a, b: Field[float]):
        with computation(PARALLEL), interval(...):
            b = a

    refused(unannotated, GTScriptDefinitionError)

    def default_before_field(a: Field[float], *, alpha: float = 0.5):
        with computation(PARALLEL), interval(...):
            a = alpha * a

    st = parse(default_before_field)
    assert [p.name for p in st.params] == ["alpha"] and np.dtype(st.params[0].dtype) == np.dtype(np.float64)

    def string_dtype(a: Field["my_real"], b: Field["my_real"]):  # noqa: F821
        with computation(PARALLEL), interval(...):
            b = a

    st = parse(string_dtype, dtypes={"my_real": np.float32})
    assert {np.dtype(f.dtype) for f in st.fields} == {np.dtype(np.float32)}
    refused(string_dtype, (GTScriptDefinitionError, GTScriptSymbolError, ValueError, KeyError))  # the name is bound nowhere


def test_lower_dimensional_and_data_dimension_fields():
    def surface_flux(flux: Field[IJ, float], rho: Field[K, float], q: Field[float]):
        with computation(FORWARD), interval(0, 1):
            q = flux * rho

    st = parse(surface_flux)
    axes = {f.name: tuple(f.axes) for f in st.fields}
    assert axes == {"flux": ("I", "J"), "rho": ("K",), "q": ("I", "J", "K")}

    def tracers(c: Field[IJK, (float, (3,))], total: Field[float]):
        with computation(PARALLEL), interval(...):
            total = c[0, 0, 0][0] + c[0, 0, 0][1] + c[1, 0, 0][2]

    st = parse(tracers)
    (c,) = [f for f in st.fields if f.name == "c"]
    assert tuple(c.data_dims) == (3,)

    def data_index_out_of_range(c: Field[IJK, (float, (3,))], total: Field[float]):
        with computation(PARALLEL), interval(...):
            total = c[0, 0, 0][3]

    refused(data_index_out_of_range, GTScriptSyntaxError)


# ---- native functions and operators (visit_Call :1733-1757; NativeFunction table nodes.py) ----------------------------------------------------
def test_native_functions_and_operator_nodes():
    def saturation(t: Field[float], es: Field[float]):
        with computation(PARALLEL), interval(...):
            es = 610.78 * exp(17.27 * (t - 273.16) / (t - 35.86)) + sqrt(abs(t)) + log(max(t, 1.0)) + min(t, 0.0) ** 2

    (stmt,) = assignments(parse(saturation))
    called = {n.func for n in ir.walk(stmt.value) if isinstance(n, ir.NativeCall)}
    assert {"exp", "sqrt", "abs", "log", "max", "min"} <= called

    def wrong_arity(t: Field[float], es: Field[float]):
        with computation(PARALLEL), interval(...):
            es = exp(t, t)

    refused(wrong_arity, GTScriptSyntaxError)

    def two_comparisons(t: Field[float], m: Field[float]):
        with computation(PARALLEL), interval(...):
            m = 1.0 if (250.0 < t) and (t < 300.0) else 0.0

    (stmt,) = assignments(parse(two_comparisons))  # BoolOp -> a left-leaning tree of binary `and` (visit_BoolOp :1551-1564)
    cond = stmt.value.cond
    assert isinstance(cond, ir.BinaryOp) and cond.op == "and" and {cond.left.op, cond.right.op} == {"<"}


def test_literals_follow_the_literal_precision_options():
    def constants(a: Field[np.float32]):
        with computation(PARALLEL), interval(...):
            a = a * 0.5 + 2

    (s64,) = assignments(parse(constants))
    (s32,) = assignments(parse(constants, literal_float_precision=32, literal_int_precision=32))
    assert {np.dtype(e.dtype) for e in ir.walk(s64.value) if isinstance(e, ir.Literal)} == {np.dtype(np.float64), np.dtype(np.int64)}
    assert {np.dtype(e.dtype) for e in ir.walk(s32.value) if isinstance(e, ir.Literal)} == {np.dtype(np.float32), np.dtype(np.int32)}


# ---- enumerations and absolute K indexing ------------------------------------------------------------------------------------------------------
class Scheme(enum.IntEnum):
    UPWIND = 1
    CENTRED = 2


def test_int_enum_members_are_integer_constants():
    def pick(a: Field[float]):
        with computation(PARALLEL), interval(...):
            a = 1.0 if Scheme.CENTRED == 2 else 0.0

    (stmt,) = assignments(parse(pick))
    assert 2 in literal_values(stmt.value)


def test_absolute_k_index():
    def from_surface(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b = a.at(K=0) + a

    (stmt,) = assignments(parse(from_surface))
    assert any(getattr(n, "absolute_k", None) is not None or type(n).__name__ == "AbsoluteKAccess" for n in ir.walk(stmt.value))

    def absolute_horizontal(a: Field[float], b: Field[float]):
        with computation(PARALLEL), interval(...):
            b = a.at(I=0)

    refused(absolute_horizontal, GTScriptSyntaxError)
