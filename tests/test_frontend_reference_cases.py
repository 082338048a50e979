"""Accept / reject behaviour of the GTScript frontend, case by case as the reference's own unit tests pin it.

Restates /root/reference/tests/cartesian_tests/unit_tests/frontend_tests/test_gtscript_frontend.py (class and line
cited per section): the same small definitions, the same exception types and message patterns.  Assertions on the
reference's internal definition-IR node classes are replaced by the equivalent facts of this repo's IR.
"""

import dataclasses
import types
import zlib
from enum import IntEnum

import numpy as np
import pytest

from gt4py_amd.cartesian import definitions as D, frontend, gtscript, ir
from gt4py_amd.cartesian.definitions import GTScriptDefinitionError, GTScriptSymbolError, GTScriptSyntaxError
from gt4py_amd.cartesian.gtscript import (  # noqa: F401
    FORWARD, BACKWARD, IJ, IJK, PARALLEL, Field, I, J, K, abs, asin, compile_assert, computation, horizontal, interval,
    isfinite, region, sin, float32, float64, int32, int64,
)


def parse_definition(definition, *, externals=None, dtypes=None, literal_int_precision=None, literal_float_precision=None):
    kw = {}
    if literal_int_precision is not None:
        kw["literal_int_precision"] = literal_int_precision
    if literal_float_precision is not None:
        kw["literal_float_precision"] = literal_float_precision
    options = D.BuildOptions(name=definition.__name__, module=__name__, backend_opts={}, **kw)
    return frontend.parse_stencil(definition, externals=externals or {}, dtypes=dtypes or {}, options=options)


def statements(stencil):
    return [s for _, _, s in stencil.statements()]


def literals(expr):
    return [e for e in ir.walk(expr) if isinstance(e, ir.Literal)]


GLOBAL_BOOL_CONSTANT = True
GLOBAL_CONSTANT = 1.0
GLOBAL_CONSTANT_I32 = np.int32(1)
GLOBAL_CONSTANT_I64 = np.int64(1)
GLOBAL_CONSTANT_F32 = np.float32(1.0)
GLOBAL_CONSTANT_F64 = np.float64(1.0)
GLOBAL_NESTED_CONSTANTS = types.SimpleNamespace(A=100, B=200)
GLOBAL_VERY_NESTED_CONSTANTS = types.SimpleNamespace(nested=types.SimpleNamespace(A=1000, B=2000))


class GlobalConstants:
    i32 = np.int32(1)
    i64 = np.int64(1)
    f32 = np.float32(1.0)
    f64 = np.float64(1.0)


@dataclasses.dataclass
class GlobalConstantsDataclass:
    i32: np.int32 = np.int32(1)
    i64: np.int64 = np.int64(1)
    f32: np.float32 = np.float32(1.0)
    f64: np.float64 = np.float64(1.0)


@gtscript.function
def add_external_const(a):
    return a + 10.0 + GLOBAL_CONSTANT


# ---- TestInlinedExternals (:126-315) ---------------------------------------------------------------------
class TestInlinedExternals:
    def test_all_legal_combinations(self):
        def definition_func(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = ((inout_field[0, 0, 0] + GLOBAL_CONSTANT + GLOBAL_NESTED_CONSTANTS.A
                                + GLOBAL_VERY_NESTED_CONSTANTS.nested.A) if GLOBAL_BOOL_CONSTANT else 0)

        (stmt,) = statements(parse_definition(definition_func))
        assert {lit.value for lit in literals(stmt.value)} >= {1.0, 100, 1000}

    def test_typed_globals(self):
        def my_stencil(field: Field[float]):
            with computation(PARALLEL), interval(...):
                i32 = GLOBAL_CONSTANT_I32
                i64 = GLOBAL_CONSTANT_I64
                f32 = GLOBAL_CONSTANT_F32
                f64 = GLOBAL_CONSTANT_F64
                c_i32 = GlobalConstants.i32
                c_i64 = GlobalConstants.i64
                c_f32 = GlobalConstants.f32
                c_f64 = GlobalConstants.f64
                dc_i32 = GlobalConstantsDataclass.i32
                dc_i64 = GlobalConstantsDataclass.i64
                dc_f32 = GlobalConstantsDataclass.f32
                dc_f64 = GlobalConstantsDataclass.f64
                field = i32 + c_i32 + dc_i32
                field = i64 + c_i64 + dc_i64
                field = f32 + c_f32 + dc_f32
                field = f64 + c_f64 + dc_f64

        stmts = statements(parse_definition(my_stencil, literal_float_precision=32, literal_int_precision=32))
        # typed numpy constants keep their precision whatever the literal precision is
        want = [np.int32, np.int64, np.float32, np.float64] * 3
        assert [s.value.dtype for s in stmts[:12]] == [np.dtype(t) for t in want]

    def test_missing(self):
        def definition_func(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = inout_field[0, 0, 0] + MISSING_CONSTANT  # noqa: F821

        with pytest.raises(GTScriptSymbolError, match=r".*MISSING_CONSTANT.*"):
            parse_definition(definition_func)

        def definition_func(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = inout_field[0, 0, 0] + GLOBAL_NESTED_CONSTANTS.missing

        with pytest.raises(GTScriptDefinitionError, match=r".*GLOBAL_NESTED_CONSTANTS.missing.*"):
            parse_definition(definition_func)

    def test_recursive_function_imports(self):
        @gtscript.function
        def func_deeply_nested():
            from gt4py.cartesian.__externals__ import another_const

            return another_const

        @gtscript.function
        def func_nested():
            from gt4py.cartesian.__externals__ import const

            return const + func_deeply_nested()

        @gtscript.function
        def func():
            from gt4py.cartesian.__externals__ import other_call

            return other_call()

        def definition_func(inout_field: Field[float]):
            from gt4py.cartesian.__externals__ import some_call

            with computation(PARALLEL), interval(...):
                inout_field = func() + some_call()

        (stmt,) = [s for s in statements(parse_definition(definition_func, externals={
            "some_call": func, "other_call": func_nested, "const": GLOBAL_CONSTANT, "another_const": GLOBAL_CONSTANT}))
            if s.target.name == "inout_field"]
        assert stmt is not None

    def test_decorated_freeze(self):
        A = 0

        @gtscript.function
        def some_function():
            return A

        A = 1  # noqa: F841 - the function saw A == 0 when it was decorated

        def definition_func(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = some_function()

        stmts = statements(parse_definition(definition_func, externals={"func": some_function}))
        values = [lit.value for s in stmts for lit in literals(s.value)]
        assert values == [0]

    @pytest.mark.parametrize("value_type", [str, dict, list])
    def test_wrong_value(self, value_type):
        WRONG_VALUE_CONSTANT = value_type()

        def definition_func(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = inout_field[0, 0, 0] + WRONG_VALUE_CONSTANT

        with pytest.raises(GTScriptDefinitionError, match=r".*WRONG_VALUE_CONSTANT.*"):
            parse_definition(definition_func)

    def test_np_bool_external(self):
        def stencil(input_field: Field[float], output_field: Field[float]):
            from __externals__ import flag

            with computation(PARALLEL), interval(...):
                add_me = 1 if flag else 5
                output_field = input_field + add_me

        parse_definition(stencil, externals={"flag": np.bool_(True)})


# ---- TestFunction (:318-483) -------------------------------------------------------------------------------
class TestFunction:
    def test_error_invalid(self):
        def func():
            return 1.0

        def definition_func(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = func()

        with pytest.raises(TypeError, match=r"func is not a gtscript function"):
            parse_definition(definition_func)

    def test_use_in_expressions_and_arguments(self):
        @gtscript.function
        def func():
            return 1.0

        @gtscript.function
        def func_outer(arg):
            return arg + 1

        def in_expr(inout_field: Field[float]):
            from gt4py.cartesian.__gtscript__ import PARALLEL, computation, interval

            with computation(PARALLEL), interval(...):
                inout_field = func() + 1

        def as_arg(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = func_outer(func())

        def expr_in_arg(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = func_outer(func() + 1)

        for definition in (in_expr, as_arg, expr_in_arg):
            parse_definition(definition)

    def test_multiple_return_values_in_expressions(self):
        @gtscript.function
        def func():
            tmp1 = 1
            tmp2 = 2
            return tmp1, tmp2

        @gtscript.function
        def func_outer(arg):
            return arg + 1

        def in_expr(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = func() + 1

        def as_arg(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = func_outer(func())

        for definition in (in_expr, as_arg):
            with pytest.raises(GTScriptSyntaxError, match="Only functions with a single return value can be used in "
                                                          "expressions, including as call arguments. Please assign the "
                                                          "function results to symbols first."):
                parse_definition(definition)

    def test_recursive_function_call_two_externals(self):
        @gtscript.function
        def func1(arg):
            from __externals__ import func2

            return func2(arg)

        @gtscript.function
        def func2(arg):
            from __externals__ import func1

            return func1(arg)

        def definition_func(inout_field: Field[float]):
            with computation(PARALLEL), interval(...):
                inout_field = func2(inout_field)

        with pytest.raises(GTScriptSyntaxError, match="recursive function call"):
            parse_definition(definition_func, externals={"func1": func1, "func2": func2})

    def test_recursive_function_calls_external_self(self):
        @gtscript.function
        def recursive_fcn(arg):
            from gt4py.cartesian.__externals__ import func

            return func(arg + 1)

        def definition_func(phi: Field[np.float64]):
            with computation(PARALLEL), interval(...):
                phi = recursive_fcn(phi)

        with pytest.raises(GTScriptSyntaxError, match="recursive"):
            parse_definition(definition_func, externals={"func": recursive_fcn})


# ---- TestLazyFunction (:486-535) ---------------------------------------------------------------------------
class TestLazyFunction:
    def test_simple_case(self):
        @gtscript.lazy_function()
        def constant():
            return 1.0

        def definition_func(out_field: Field[float]):
            with computation(PARALLEL), interval(...):
                out_field = constant()

        stmts = statements(parse_definition(definition_func))
        assert [lit.value for s in stmts for lit in literals(s.value)] == [1.0]

    def test_annotation_deferred_type(self):
        class DeferredType:
            pass

        def resolve_type(func):
            for name, type_ in func.__annotations__.items():
                if type_ == DeferredType:
                    func.__annotations__[name] = Field[float]

        @gtscript.lazy_function(before_annotation=resolve_type)
        def plus_one(a: DeferredType):
            return a + 1.0

        def definition_func(in_field: DeferredType, out_field: Field[float]):
            with computation(PARALLEL), interval(...):
                out_field = plus_one(in_field)

        resolve_type(definition_func)
        stmts = statements(parse_definition(definition_func))
        assert 1.0 in [lit.value for s in stmts for lit in literals(s.value)]


# ---- TestAxisSyntax (:538-623) -----------------------------------------------------------------------------
class TestAxisSyntax:
    def test_good_syntax(self):
        def definition_func(in_field: Field[float], out_field: Field[float]):
            with computation(PARALLEL), interval(...):
                out_field = in_field[J - 1] + in_field[J]

        (stmt,) = statements(parse_definition(definition_func))
        assert sorted(e.offset for e in ir.walk(stmt.value) if isinstance(e, ir.FieldAccess)) == [(0, -1, 0), (0, 0, 0)]

    def test_good_syntax_external(self):
        def definition_func(in_field: Field[float], out_field: Field[float]):
            from gt4py.cartesian.__externals__ import AXIS

            with computation(PARALLEL), interval(...):
                out_field = in_field[AXIS - 1]

        (stmt,) = statements(parse_definition(definition_func, externals={"AXIS": gtscript.Axis("I")}))
        assert [e.offset for e in ir.walk(stmt.value) if isinstance(e, ir.FieldAccess)] == [(-1, 0, 0)]

    def test_good_syntax_external_value(self):
        def definition_func(in_field: Field[float], out_field: Field[float]):
            from gt4py.cartesian.__externals__ import VALUE

            with computation(PARALLEL), interval(...):
                out_field = in_field[J - VALUE]

        for value in range(2):
            (stmt,) = statements(parse_definition(definition_func, externals={"VALUE": value}))
            assert [e.offset for e in ir.walk(stmt.value) if isinstance(e, ir.FieldAccess)] == [(0, -value, 0)]

    @pytest.mark.parametrize("index", ["I * 1", "I + 1 + I", "I, I - 1", "J, I - 1"])
    def test_bad_syntax(self, index):
        ns = {}
        src = ("def definition_func(in_field: Field[float], out_field: Field[float]):\n"
               "    with computation(PARALLEL), interval(...):\n"
               f"        out_field = in_field[{index}]\n")
        definition = _compile(src, "definition_func", ns)
        with pytest.raises(GTScriptSyntaxError):
            parse_definition(definition)


def _compile(src, name, ns=None):
    """Function object with retrievable source (GTScript definitions are read with ``inspect``)."""
    import linecache

    ns = dict(globals()) if ns is None else {**globals(), **ns}
    filename = f"<frontend_case_{zlib.crc32(src.encode())}>"
    linecache.cache[filename] = (len(src), None, src.splitlines(True), filename)
    exec(compile(src, filename, "exec"), ns)  # noqa: S102 - test-local source
    return ns[name]


# ---- TestImportedExternals (:626-712) ----------------------------------------------------------------------
class TestImportedExternals:
    def test_all_legal_combinations(self):
        externals = dict(BOOL_CONSTANT=-1.0, CONSTANT=-2.0, NESTED_CONSTANTS=types.SimpleNamespace(A=-100, B=-200),
                         VERY_NESTED_CONSTANTS=types.SimpleNamespace(nested=types.SimpleNamespace(A=-1000, B=-2000)))

        def definition_func(inout_field: Field[float]):
            from gt4py.cartesian.__externals__ import BOOL_CONSTANT, CONSTANT, NESTED_CONSTANTS, VERY_NESTED_CONSTANTS

            with computation(PARALLEL), interval(...):
                inout_field = ((inout_field[0, 0, 0] + CONSTANT + NESTED_CONSTANTS.A + VERY_NESTED_CONSTANTS.nested.A)
                               if GLOBAL_BOOL_CONSTANT else 0)

        (stmt,) = statements(parse_definition(definition_func, externals=externals))
        assert {lit.value for lit in literals(stmt.value)} >= {-2.0, -100, -1000}

    def test_missing(self):
        externals = dict(CONSTANT=-2.0, NESTED_CONSTANTS=types.SimpleNamespace(A=-100, B=-200))

        def definition_func(inout_field: Field[float]):
            from gt4py.cartesian.__externals__ import MISSING_CONSTANT

            with computation(PARALLEL), interval(...):
                inout_field = inout_field[0, 0, 0] + MISSING_CONSTANT

        with pytest.raises(GTScriptDefinitionError, match=r".*MISSING_CONSTANT.*"):
            parse_definition(definition_func)

        def definition_func(inout_field: Field[float]):
            from gt4py.cartesian.__externals__ import NESTED_CONSTANTS

            with computation(PARALLEL), interval(...):
                inout_field = inout_field[0, 0, 0] + NESTED_CONSTANTS.missing

        with pytest.raises(GTScriptDefinitionError, match=r".*NESTED_CONSTANTS.missing.*"):
            parse_definition(definition_func, externals=externals)

    @pytest.mark.parametrize("value_type", [str, dict, list])
    def test_wrong_value(self, value_type):
        def definition_func(inout_field: Field[float]):
            from gt4py.cartesian.__externals__ import WRONG_VALUE_CONSTANT

            with computation(PARALLEL), interval(...):
                inout_field = inout_field[0, 0, 0] + WRONG_VALUE_CONSTANT

        with pytest.raises(GTScriptDefinitionError, match=r".*WRONG_VALUE_CONSTANT.*"):
            parse_definition(definition_func, externals=dict(WRONG_VALUE_CONSTANT=value_type()))
