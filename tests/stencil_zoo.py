"""Stencil definitions shared by the code-generation tests (CPU: planning + compilation; GPU: values).

Most are restatements of definitions in the reference's test-suite
(/root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/stencil_definitions.py and
test_suites.py); the rest are small shapes that pin one decision of the stage planner each.
Annotation strings are evaluated by ``gtscript.stencil`` (Field, IJ, K, np are in scope there).
"""
# flake8: noqa: F821, F841
import numpy as np

from gt4py_amd.cartesian.gtscript import Field  # noqa: F401 - used by annotations inside definitions

F64 = "Field[np.float64]"
F32 = "Field[np.float32]"


def copy_stencil(a: F64, b: F64):
    """stencil_definitions.py:66-69"""
    with computation(PARALLEL), interval(...):
        b = a


def native_functions(a: F64, b: F64):
    """stencil_definitions.py:145-153 (the exactly-rounded subset)"""
    with computation(PARALLEL), interval(...):
        abs_res = abs(a)
        max_res = max(abs_res, b)
        min_res = min(max_res, 42)
        b = sqrt(abs(min_res)) + floor(a) - ceil(b) + trunc(a * 3.5) + (a % 0.3)


def ternary_mix(a: F64, b: F64, c: F64):
    with computation(PARALLEL), interval(...):
        c = max(a, b) if (a > 0.0 and b < 0.5) or not (a < -0.5) else sqrt(abs(a)) / (b + 2.0)


def mixed_precision(a: F32, b: F64, c: F32, *, w: np.float32):
    """f32 fields with f64 literals: everything computed in f64, rounded once on store."""
    with computation(PARALLEL), interval(...):
        t = a * 0.3 + b
        c = t * w - a / 3.0


def int_fields(a: "Field[np.int32]", b: "Field[np.int64]", c: "Field[np.int64]"):
    with computation(PARALLEL), interval(...):
        c = a * 3 + b - (a if a > b else b) + (a % 7)


def k_intervals(a: F64, b: F64, out: F64):
    """different expressions per K interval, reads of inputs at K offsets (large_k_interval-like)"""
    with computation(PARALLEL):
        with interval(0, 2):
            out = a
        with interval(2, -1):
            out = a + b[0, 0, -2] * b[0, 0, 1]
        with interval(-1, None):
            out = a - b[0, 0, -1]


def horizontal_diffusion(in_field: F64, out_field: F64, coeff: F64):
    """stencil_definitions.py:316-328"""
    with computation(PARALLEL), interval(...):
        lap_field = 4.0 * in_field[0, 0, 0] - (
            in_field[1, 0, 0] + in_field[-1, 0, 0] + in_field[0, 1, 0] + in_field[0, -1, 0]
        )
        res = lap_field[1, 0, 0] - lap_field[0, 0, 0]
        flx_field = 0 if (res * (in_field[1, 0, 0] - in_field[0, 0, 0])) > 0 else res
        res = lap_field[0, 1, 0] - lap_field[0, 0, 0]
        fly_field = 0 if (res * (in_field[0, 1, 0] - in_field[0, 0, 0])) > 0 else res
        out_field = in_field[0, 0, 0] - coeff[0, 0, 0] * (
            flx_field[0, 0, 0] - flx_field[-1, 0, 0] + fly_field[0, 0, 0] - fly_field[0, -1, 0]
        )


def horizontal_diffusion_f32(in_field: F32, out_field: F32, coeff: F32):
    with computation(PARALLEL), interval(...):
        lap_field = 4.0 * in_field[0, 0, 0] - (
            in_field[1, 0, 0] + in_field[-1, 0, 0] + in_field[0, 1, 0] + in_field[0, -1, 0]
        )
        res = lap_field[1, 0, 0] - lap_field[0, 0, 0]
        flx_field = 0 if (res * (in_field[1, 0, 0] - in_field[0, 0, 0])) > 0 else res
        res = lap_field[0, 1, 0] - lap_field[0, 0, 0]
        fly_field = 0 if (res * (in_field[0, 1, 0] - in_field[0, 0, 0])) > 0 else res
        out_field = in_field[0, 0, 0] - coeff[0, 0, 0] * (
            flx_field[0, 0, 0] - flx_field[-1, 0, 0] + fly_field[0, 0, 0] - fly_field[0, -1, 0]
        )


def laplacian(inp: F64, out: F64):
    with computation(PARALLEL), interval(...):
        out = -4.0 * inp[0, 0, 0] + inp[-1, 0, 0] + inp[1, 0, 0] + inp[0, -1, 0] + inp[0, 1, 0]


def horizontal_diffusion_if(in_field: F64, out_field: F64, coeff: F64):
    """The flux-limited horizontal diffusion with its limiter written as `if` / `else` blocks (the way hand-ported Fortran
    operators are) instead of the conditional expressions of stencil_definitions.py:316-328: same values."""
    with computation(PARALLEL), interval(...):
        lap_field = 4.0 * in_field[0, 0, 0] - (in_field[1, 0, 0] + in_field[-1, 0, 0] + in_field[0, 1, 0] + in_field[0, -1, 0])
        res = lap_field[1, 0, 0] - lap_field[0, 0, 0]
        if res * (in_field[1, 0, 0] - in_field[0, 0, 0]) > 0:
            flx_field = 0.0
        else:
            flx_field = res
        res = lap_field[0, 1, 0] - lap_field[0, 0, 0]
        if res * (in_field[0, 1, 0] - in_field[0, 0, 0]) > 0:
            fly_field = 0.0
        else:
            fly_field = res
        out_field = in_field[0, 0, 0] - coeff[0, 0, 0] * (
            flx_field[0, 0, 0] - flx_field[-1, 0, 0] + fly_field[0, 0, 0] - fly_field[0, -1, 0])


def hyperdiffusion_6th(inp: F64, out: F64, *, nu: float):
    """Laplacian applied three times (a sixth-order filter), with a boundary level of its own that uses the same
    temporaries: the inputs are read up to 3 points away through a chain of temporaries (two halo lanes in the strip
    kernel with shared temporaries), and `l1` / `l2` are one temporary per interval block."""
    with computation(PARALLEL):
        with interval(0, 1):
            l1 = -4.0 * inp + inp[1, 0, 0] + inp[-1, 0, 0] + inp[0, 1, 0] + inp[0, -1, 0]
            l2 = -4.0 * l1 + l1[1, 0, 0] + l1[-1, 0, 0] + l1[0, 1, 0] + l1[0, -1, 0]
            out = inp + 0.5 * nu * l2
        with interval(1, None):
            l1 = -4.0 * inp + inp[1, 0, 0] + inp[-1, 0, 0] + inp[0, 1, 0] + inp[0, -1, 0]
            l2 = -4.0 * l1 + l1[1, 0, 0] + l1[-1, 0, 0] + l1[0, 1, 0] + l1[0, -1, 0]
            l3 = -4.0 * l2 + l2[1, 0, 0] + l2[-1, 0, 0] + l2[0, 1, 0] + l2[0, -1, 0]
            out = inp + nu * l3


def tridiagonal_solver(inf: F64, diag: F64, sup: F64, rhs: F64, out: F64):
    """stencil_definitions.py:219-232"""
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


def vertical_advection_dycore(utens_stage: F64, u_stage: F64, wcon: F64, u_pos: F64, utens: F64, *, dtr_stage: float):
    """stencil_definitions.py:235-313"""
    from __externals__ import BET_M, BET_P

    with computation(FORWARD):
        with interval(0, 1):
            gcv = 0.25 * (wcon[1, 0, 1] + wcon[0, 0, 1])
            cs = gcv * BET_M

            ccol = gcv * BET_P
            bcol = dtr_stage - ccol[0, 0, 0]

            # update the d column
            correction_term = -cs * (u_stage[0, 0, 1] - u_stage[0, 0, 0])
            dcol = dtr_stage * u_pos[0, 0, 0] + utens[0, 0, 0] + utens_stage[0, 0, 0] + correction_term

            # Thomas forward
            divided = 1.0 / bcol[0, 0, 0]
            ccol = ccol[0, 0, 0] * divided
            dcol = dcol[0, 0, 0] * divided

        with interval(1, -1):
            gav = -0.25 * (wcon[1, 0, 0] + wcon[0, 0, 0])
            gcv = 0.25 * (wcon[1, 0, 1] + wcon[0, 0, 1])

            as_ = gav * BET_M
            cs = gcv * BET_M

            acol = gav * BET_P
            ccol = gcv * BET_P
            bcol = dtr_stage - acol[0, 0, 0] - ccol[0, 0, 0]

            # update the d column
            correction_term = -as_ * (u_stage[0, 0, -1] - u_stage[0, 0, 0]) - cs * (
                u_stage[0, 0, 1] - u_stage[0, 0, 0]
            )
            dcol = dtr_stage * u_pos[0, 0, 0] + utens[0, 0, 0] + utens_stage[0, 0, 0] + correction_term

            # Thomas forward
            divided = 1.0 / (bcol[0, 0, 0] - ccol[0, 0, -1] * acol[0, 0, 0])
            ccol = ccol[0, 0, 0] * divided
            dcol = (dcol[0, 0, 0] - (dcol[0, 0, -1]) * acol[0, 0, 0]) * divided

        with interval(-1, None):
            gav = -0.25 * (wcon[1, 0, 0] + wcon[0, 0, 0])
            as_ = gav * BET_M
            acol = gav * BET_P
            bcol = dtr_stage - acol[0, 0, 0]

            # update the d column
            correction_term = -as_ * (u_stage[0, 0, -1] - u_stage[0, 0, 0])
            dcol = dtr_stage * u_pos[0, 0, 0] + utens[0, 0, 0] + utens_stage[0, 0, 0] + correction_term

            # Thomas forward
            divided = 1.0 / (bcol[0, 0, 0] - ccol[0, 0, -1] * acol[0, 0, 0])
            dcol = (dcol[0, 0, 0] - (dcol[0, 0, -1]) * acol[0, 0, 0]) * divided

    with computation(BACKWARD):
        with interval(-1, None):
            datacol = dcol[0, 0, 0]
            utens_stage = dtr_stage * (datacol - u_pos[0, 0, 0])

        with interval(0, -1):
            datacol = dcol[0, 0, 0] - ccol[0, 0, 0] * datacol[0, 0, 1]
            utens_stage = dtr_stage * (datacol - u_pos[0, 0, 0])


def column_sum_then_gradient(a: F64, out: F64):
    """a FORWARD-accumulated temporary read at horizontal offsets afterwards: two stages, scratch"""
    with computation(FORWARD):
        with interval(0, 1):
            acc = a
        with interval(1, None):
            acc = acc[0, 0, -1] + a
    with computation(PARALLEL), interval(...):
        out = acc[1, 0, 0] - acc[-1, 0, 0] + acc[0, 1, 0]


def backward_scan(a: F64, out: F64):
    """test_code_generation.py:622-649 shape: BACKWARD recurrence over an API field"""
    with computation(BACKWARD):
        with interval(-1, None):
            out = a + 1.0
        with interval(0, -1):
            out = out[0, 0, 1] * 0.5 + a[0, 0, 1]


def parallel_k_dependency(a: F64, b: F64):
    """PARALLEL computations that read a field at a K offset, overwrite it, and read the new values at an
    offset again (one computation each: within ONE parallel loop the reference forbids it, gtir.py:243-293)"""
    with computation(PARALLEL), interval(0, -1):
        tmp = a[0, 0, 1] * 2.0
    with computation(PARALLEL), interval(0, -1):
        a = tmp + b
    with computation(PARALLEL), interval(0, -2):
        b = a[0, 0, 1] - tmp


def temporary_read_at_k_offset_in_later_loop(a: F64, out: F64):
    """A temporary written in one PARALLEL loop and read at a K offset in a LATER one: legal in the reference
    (gtir.py:243-293 only looks inside one VerticalLoop; K boundary 0 for both, gtir_k_boundary.py:39-70)."""
    with computation(PARALLEL), interval(...):
        tmp = a * 2.0
    with computation(PARALLEL), interval(0, -1):
        out = tmp[0, 0, 1] - tmp
    with computation(PARALLEL), interval(-1, None):
        out = tmp[0, 0, -1]


def two_sweep_three_carried(a: F64, w: F32, x: F64, y: F32, out: F64, *, alpha: float):
    """A forward recurrence over three values -- a float64 temporary, a float32 temporary and the float64 API field `x`
    (in/out: its old value is read at the level before it is overwritten) -- that the backward sweep reads back level by
    level: the top of the column stays in registers + LDS between the sweeps (stage_planner.TopCache, mixed item sizes,
    one store-through field, five interval blocks)."""
    with computation(FORWARD):
        with interval(0, 2):
            p = a * alpha
            q = w
            x = x + p
        with interval(2, -3):
            p = a * alpha + p[0, 0, -1] * 0.5
            q = w - q[0, 0, -1] * 0.25
            x = x * 0.5 + p - x[0, 0, -1] * 0.125
        with interval(-3, None):
            p = p[0, 0, -1] - a
            q = q[0, 0, -1] * w
            x = x[0, 0, -1] + p
    with computation(BACKWARD):
        with interval(-1, None):
            out = p + q + x
            y = q
        with interval(0, -1):
            out = p * out[0, 0, 1] + q - x
            y = q + w if p > 0.0 else q - w


def cross_column_recurrence(a: F64, b: F64, c: F64):
    """Sequential blocks whose columns depend on each other through a temporary read at horizontal offsets
    (legal in the reference: gtir.py:224-241 only forbids it for API fields): forward, then backward with a mask."""
    with computation(FORWARD), interval(1, None):
        t = b[0, 0, -1] * 0.5 + b
        a = a[0, 0, -1] * 0.5 + t[1, 0, 0] * 0.5 + t[-1, 0, 0] * 0.25
    with computation(BACKWARD):
        with interval(-1, None):
            c = a
        with interval(0, -1):
            u = b[0, 0, 1] - b
            if u[0, 1, 0] > 0.0:
                c = u[0, -1, 0] + c[0, 0, 1]
            else:
                c = u + a


def plane_recurrence(b: F64, out: F64):
    """A vertical recurrence over a TEMPORARY that the same block reads at horizontal offsets: nothing can be taken
    out of the loop, the block runs plane by plane (legal in the reference: only temporaries cross columns)."""
    with computation(FORWARD):
        with interval(0, 1):
            s = b
            out = s
        with interval(1, None):
            s = s[0, 0, -1] * 0.5 + b
            t = s * 2.0
            out = t[1, 0, 0] + s[-1, 0, 0] + out[0, 0, -1] * 0.25


def lower_dimensional(a: F64, surf: "Field[IJ, np.float64]", prof: "Field[K, np.float64]", out: F64):
    """test_code_generation.py:178-314 shape: IJ and K fields broadcast against a 3-d field"""
    with computation(PARALLEL), interval(...):
        out = a * surf + prof[1] * surf[1, 0] - prof


def two_stage_written_input(a: F64, b: F64, out: F64):
    """a temporary that depends on a *written* API field cannot be recomputed: scratch + second stage"""
    with computation(PARALLEL), interval(...):
        b = a * 2.0
        t = b + a
        out = t[1, 0, 0] + t[0, -1, 0]


def runtime_if(field_a: F64, field_b: F64):
    """stencil_definitions.py:195-203"""
    with computation(BACKWARD), interval(...):
        if field_a > 0.0:
            field_b = -1
            field_a = -field_a
        else:
            field_b = 1
            field_a = field_a


def nested_if(a: F64, b: F64, out: F64, *, thresh: float):
    """field and scalar conditions, nesting, a temporary assigned in both branches, a recurrence"""
    with computation(FORWARD):
        with interval(0, 1):
            out = a
        with interval(1, None):
            if a > b:
                t = a - b
                if thresh > 0.0:
                    t = t * thresh
                    if b < 0.0:
                        t = -t
            else:
                t = out[0, 0, -1]
            out = t + out[0, 0, -1] * 0.5


def if_with_offsets(a: F64, out: F64):
    """condition and body read at horizontal offsets; the whole `if` is one horizontal execution"""
    with computation(PARALLEL), interval(...):
        lap = a[1, 0, 0] + a[-1, 0, 0] - 2.0 * a
        if lap[0, 1, 0] > lap[0, -1, 0]:
            out = lap[1, 0, 0]
        else:
            out = a[0, 1, 0] if lap > 0.0 else a[0, -1, 0]


def variable_k_offsets(a: F64, idx: "Field[np.int32]", out: F64):
    """run-time K index from a 3-d integer field, also read at a horizontal offset (VariableKOffset)"""
    with computation(PARALLEL), interval(1, -1):
        t = a[1, 0, idx[1, 0, 0]] - a[0, 0, idx]
        out = t[-1, 0, 0] + a[0, 0, idx + 0]


def variable_k_of_written_field(a: F64, idx: "Field[np.int32]", out: F64):
    """a field written in the stencil read back at a run-time K index: column stage, statement by statement"""
    with computation(PARALLEL), interval(...):
        out = a * 2.0
    with computation(PARALLEL), interval(1, -1):
        a = out[0, 0, idx] + 1.0


def newton_sqrt(a: F64, out: F64, *, tol: float):
    """data-dependent iteration count per point: Newton's method inside a run-time `while`"""
    with computation(PARALLEL), interval(...):
        x = 1.0
        target = abs(a) + 0.5
        while abs(x * x - target) > tol:
            x = 0.5 * (x + target / x)
        out = x


def while_in_if_and_scan(a: F64, out: F64):
    """`while` nested in `if`/`else` inside a FORWARD recurrence, with a counter that bounds the loop"""
    with computation(FORWARD):
        with interval(0, 1):
            out = a
        with interval(1, None):
            v = a + out[0, 0, -1] * 0.5
            n = 0
            if v > 0.0:
                while v > 0.25 and n < 6:
                    v = v * 0.5
                    n = n + 1
            else:
                while v < -0.25:
                    v = v * 0.5 + 0.01
            out = v + n


def typed_temporary(a: F64, b: F64):
    """typed, initialised temporary (gtscript_frontend.py:2245-2263): float32 accumulator between float64 fields"""
    acc: Field[np.float32] = 1
    with computation(PARALLEL), interval(...):
        acc = acc + a
        b = acc[1, 0, 0] * 2.0 - acc


ZOO = {
    # name: (definition, externals, scalars, backend options)
    "copy_stencil": (copy_stencil, {}, {}, {}),
    "native_functions": (native_functions, {}, {}, {}),
    "cross_column_recurrence": (cross_column_recurrence, {}, {}, {}),
    "plane_recurrence": (plane_recurrence, {}, {}, {}),
    "ternary_mix": (ternary_mix, {}, {}, {}),
    "mixed_precision": (mixed_precision, {}, {"w": np.float32(0.7)}, {}),
    "int_fields": (int_fields, {}, {}, {}),
    "k_intervals": (k_intervals, {}, {}, {}),
    "horizontal_diffusion": (horizontal_diffusion, {}, {}, {"use_kernel_library": False}),
    "horizontal_diffusion_f32": (horizontal_diffusion_f32, {}, {}, {"use_kernel_library": False}),
    "laplacian": (laplacian, {}, {}, {"use_kernel_library": False}),
    "hyperdiffusion_6th": (hyperdiffusion_6th, {}, {"nu": 0.01}, {}),
    "horizontal_diffusion_if": (horizontal_diffusion_if, {}, {}, {}),
    "tridiagonal_solver": (tridiagonal_solver, {}, {}, {"use_kernel_library": False}),
    "vertical_advection_dycore": (vertical_advection_dycore, {"BET_M": 0.5, "BET_P": 0.5}, {"dtr_stage": 3.0 / 20.0}, {}),
    "column_sum_then_gradient": (column_sum_then_gradient, {}, {}, {}),
    "backward_scan": (backward_scan, {}, {}, {}),
    "parallel_k_dependency": (parallel_k_dependency, {}, {}, {}),
    "two_sweep_three_carried": (two_sweep_three_carried, {}, {"alpha": 0.375}, {}),
    "temporary_read_at_k_offset_in_later_loop": (temporary_read_at_k_offset_in_later_loop, {}, {}, {}),
    "lower_dimensional": (lower_dimensional, {}, {}, {}),
    "two_stage_written_input": (two_stage_written_input, {}, {}, {}),
    "variable_k_offsets": (variable_k_offsets, {}, {}, {}),
    "variable_k_of_written_field": (variable_k_of_written_field, {}, {}, {}),
    "newton_sqrt": (newton_sqrt, {}, {"tol": 1e-12}, {}),
    "while_in_if_and_scan": (while_in_if_and_scan, {}, {}, {}),
    "typed_temporary": (typed_temporary, {}, {}, {}),
    "runtime_if": (runtime_if, {}, {}, {}),
    "nested_if": (nested_if, {}, {"thresh": 0.75}, {}),
    "if_with_offsets": (if_with_offsets, {}, {}, {}),
}


def make_inputs(stencil_object, domain, seed=1337):
    """Random arrays sized domain + boundary for every field of a built stencil -> (arrays, origins)."""
    rng = np.random.default_rng(seed)
    arrays, origins = {}, {}
    for name, info in stencil_object.field_info.items():
        if info is None:
            continue
        shape, origin = [], []
        for axis, d in zip("IJK", domain):
            if axis not in info.axes:
                continue
            lo, hi = (max(0, int(b)) for b in info.boundary["IJK".index(axis)])
            shape.append(int(d) + lo + hi)
            origin.append(lo)
        shape += [int(n) for n in info.data_dims]  # data dimensions: whole, origin 0
        origin += [0] * len(info.data_dims)
        dt = np.dtype(info.dtype)
        if dt.kind == "f":
            data = rng.uniform(-1.0, 1.0, shape).astype(dt)
            if name == "diag":
                data = (data + 4.5).astype(dt)
        elif dt.kind in "iu":
            data = rng.integers(-1, 2, shape).astype(dt) if name == "idx" else rng.integers(-50, 50, shape).astype(dt)
        else:
            data = rng.integers(0, 2, shape).astype(dt)
        arrays[name] = data
        origins[name] = tuple(origin)
    return arrays, origins
