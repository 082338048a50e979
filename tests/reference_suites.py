"""The reference's integration suites as data: GTScript definition + independent numpy validation.

Source of every entry: /root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/
test_suites.py (line ranges in each docstring).  There a hypothesis-driven harness
(`gt4py.cartesian.testing.StencilTestSuite`) draws domains/values, runs the stencil on every backend and
compares with the suite's ``validation`` -- a plain-numpy function that does NOT go through gt4py.
Those validations are the reference's own known answers for these stencils, so they pin parity
independently of this repository's oracle.  Here each suite is one ``Suite`` record: the definition is
re-typed with explicit annotations (the reference injects dtypes through its harness), ``expected``
restates the validation on arrays laid out as ``domain + boundary``, and the table is consumed by
tests/test_reference_suites.py on the CPU (numpy oracle backend) and on the GPU (hip:mi300).
"""
# flake8: noqa: F821, F841
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Sequence, Tuple

import numpy as np

from gt4py_amd.cartesian import gtscript

F64 = "Field[np.float64]"
F32 = "Field[np.float32]"
Z = ((0, 0), (0, 0), (0, 0))


@dataclass
class Suite:
    definition: Callable
    fields: Dict[str, Tuple[Any, Tuple[Tuple[int, int], ...], Tuple[float, float]]]  # dtype, boundary, value range
    expected: Callable  # (arrays incl. boundary, params, externals, domain) -> {name: array over the domain}
    params: Dict[str, Tuple[float, float]] = field(default_factory=dict)
    externals: Sequence[Dict[str, Any]] = ({},)
    domains: Sequence[Tuple[int, int, int]] = ((1, 1, 1), (3, 4, 5), (15, 14, 13))
    optional: Dict[str, str] = field(default_factory=dict)  # field -> external that switches it on
    axes: Dict[str, str] = field(default_factory=dict)  # field -> "K", "IJ", ... (default "IJK")
    data_dims: Dict[str, Tuple[int, ...]] = field(default_factory=dict)  # field -> trailing data dimensions


def _inner(a, b):
    """View of ``a`` without its boundary ``b``."""
    return a[tuple(slice(lo, a.shape[ax] - hi) for ax, (lo, hi) in enumerate(b))]


# ---- test_suites.py:27-41 ----------------------------------------------------------------------
def identity(field_a: F64):
    with computation(PARALLEL), interval(...):
        tmp = field_a
        field_a = tmp


# ---- :44-60 ------------------------------------------------------------------------------------
def copy(field_a: F64, field_b: F64):
    with computation(PARALLEL), interval(...):
        field_b = field_a


# ---- :63-83 ------------------------------------------------------------------------------------
def aug_assign(field_a: F64, field_b: F64):
    with computation(PARALLEL), interval(...):
        field_a += 1.0
        field_a *= 2.0
        field_b -= 1.0
        field_b /= 2.0


# ---- :86-104 -----------------------------------------------------------------------------------
def global_scale(field_a: F64):
    from __externals__ import SCALE_FACTOR

    with computation(PARALLEL), interval(...):
        field_a = SCALE_FACTOR * field_a[0, 0, 0]


# ---- :107-123 ----------------------------------------------------------------------------------
def parametric_scale(field_a: F64, *, scale: float):
    with computation(PARALLEL), interval(...):
        field_a = scale * field_a


# ---- :126-170 ----------------------------------------------------------------------------------
def parametric_mix(field_a: F64, field_b: F64, field_c: F64, field_out: F32, *, weight: float, alpha_factor: float):
    from __externals__ import USE_ALPHA
    from __gtscript__ import __INLINED

    with computation(PARALLEL), interval(...):
        if __INLINED(USE_ALPHA):
            factor = alpha_factor
        else:
            factor = 1.0
        field_out = factor * field_a[0, 0, 0] - (1 - factor) * (field_b[0, 0, 0] - weight * field_c[0, 0, 0])


def _parametric_mix_expected(a, p, ext, domain):
    factor = p["alpha_factor"] if ext["USE_ALPHA"] else 1.0
    return {"field_out": (factor * a["field_a"]) - (1 - factor) * (a["field_b"] - (p["weight"] * a["field_c"]))}


# ---- :173-197 ----------------------------------------------------------------------------------
def heat_equation_ftcs(u: F64, v: F64, u_new: F64, v_new: F64, *, ru: float, rv: float):
    with computation(PARALLEL), interval(...):
        u_new = u[0, 0, 0] + ru * (u[1, 0, 0] - 2 * u[0, 0, 0] + u[-1, 0, 0])
        v_new = v[0, 0, 0] + rv * (v[0, 1, 0] - 2 * v[0, 0, 0] + v[0, -1, 0])


def _heat_expected(a, p, ext, domain):
    u, v = a["u"], a["v"]
    return {
        "u_new": u[1:-1] + p["ru"] * (u[2:] - 2 * u[1:-1] + u[:-2]),
        "v_new": v[:, 1:-1] + p["rv"] * (v[:, 2:] - 2 * v[:, 1:-1] + v[:, :-2]),
    }


# ---- :200-230 ----------------------------------------------------------------------------------
def horizontal_diffusion(u: F64, diffusion: F64, *, weight: float):
    with computation(PARALLEL), interval(...):
        laplacian = 4.0 * u[0, 0, 0] - (u[1, 0, 0] + u[-1, 0, 0] + u[0, 1, 0] + u[0, -1, 0])
        flux_i = laplacian[1, 0, 0] - laplacian[0, 0, 0]
        flux_j = laplacian[0, 1, 0] - laplacian[0, 0, 0]
        diffusion = u[0, 0, 0] - weight * (flux_i[0, 0, 0] - flux_i[-1, 0, 0] + flux_j[0, 0, 0] - flux_j[0, -1, 0])


def _hdiff_expected(a, p, ext, domain):
    u = a["u"]
    lap = 4.0 * u[1:-1, 1:-1] - (u[2:, 1:-1] + u[:-2, 1:-1] + u[1:-1, 2:] + u[1:-1, :-2])
    flux_i = lap[1:, 1:-1] - lap[:-1, 1:-1]
    flux_j = lap[1:-1, 1:] - lap[1:-1, :-1]
    return {"diffusion": u[2:-2, 2:-2] - p["weight"] * (flux_i[1:] - flux_i[:-1] + flux_j[:, 1:] - flux_j[:, :-1])}


# ---- :233-296 (subroutines; the function arrives as an external) -----------------------------------
@gtscript.function
def lap_op(u):
    """Laplacian operator."""
    return 4.0 * u[0, 0, 0] - (u[1, 0, 0] + u[-1, 0, 0] + u[0, 1, 0] + u[0, -1, 0])


@gtscript.function
def fwd_diff_op_xy(field):
    dx = field[1, 0, 0] - field[0, 0, 0]
    dy = field[0, 1, 0] - field[0, 0, 0]
    return dx, dy


@gtscript.function
def wrap1arg2return(field):
    dx, dy = fwd_diff_op_xy(field=field)
    return dx, dy


@gtscript.function
def fwd_diff_op_x(field):
    dx = field[1, 0, 0] - field[0, 0, 0]
    return dx


@gtscript.function
def fwd_diff_op_y(field):
    dy = field[0, 1, 0] - field[0, 0, 0]
    return dy


def horizontal_diffusion_subroutines(u: F64, diffusion: F64, *, weight: float):
    from __externals__ import fwd_diff

    with computation(PARALLEL), interval(...):
        laplacian = lap_op(u=u)
        flux_i, flux_j = fwd_diff(field=laplacian)
        diffusion = u[0, 0, 0] - weight * (flux_i[0, 0, 0] - flux_i[-1, 0, 0] + flux_j[0, 0, 0] - flux_j[0, -1, 0])


# ---- :299-337 ----------------------------------------------------------------------------------
def horizontal_diffusion_subroutines2(u: F64, diffusion: F64, *, weight: float):
    from __externals__ import BRANCH
    from __gtscript__ import __INLINED

    with computation(PARALLEL), interval(...):
        laplacian = lap_op(u=u)
        if __INLINED(BRANCH):
            flux_i = fwd_diff_op_x(field=laplacian)
            flux_j = fwd_diff_op_y(field=laplacian)
        else:
            flux_i, flux_j = fwd_diff_op_xy(field=laplacian)
        diffusion = u[0, 0, 0] - weight * (flux_i[0, 0, 0] - flux_i[-1, 0, 0] + flux_j[0, 0, 0] - flux_j[0, -1, 0])


# ---- :340-356 ----------------------------------------------------------------------------------
def runtime_if_flat(outfield: F64):
    with computation(PARALLEL), interval(...):
        if True:
            outfield = 1
        else:
            outfield = 2


# ---- :359-378 ----------------------------------------------------------------------------------
def runtime_if_nested(outfield: F64):
    with computation(PARALLEL), interval(...):
        if (outfield > 0 and outfield > 0) or (not outfield > 0 and not outfield > 0):
            if False:
                outfield = 1
            else:
                outfield = 2
        else:
            outfield = 3


# ---- :381-404 ----------------------------------------------------------------------------------
@gtscript.function
def add_one(field_in):
    """Add 1 to each element of `field_in`."""
    return field_in + 1


def three_fold_nested_if(field_a: F64):
    with computation(PARALLEL), interval(...):
        if field_a >= 0.0:
            field_a = 0.0
            if field_a > 1:
                field_a = 1
                if field_a > 2:
                    field_a = 2


def _three_fold_expected(a, p, ext, domain):
    out = a["field_a"].copy()
    for v in range(3):
        out[np.where(out > v)] = v
    return {"field_a": out}


# ---- :407-438 ----------------------------------------------------------------------------------
def runtime_if_nested_data_dependent(field_a: F64, field_b: F64, field_c: F64, *, factor: float):
    with computation(PARALLEL), interval(...):
        if factor > 0:
            if field_a < 0:
                field_b = -field_a
            else:
                field_b = field_a
        else:
            if field_a < 0:
                field_c = -field_a
            else:
                field_c = field_a

        field_a = add_one(field_a)


def _data_dependent_expected(a, p, ext, domain):
    out = {"field_a": a["field_a"] + 1}
    out["field_b" if p["factor"] > 0 else "field_c"] = np.abs(a["field_a"])
    return out


# ---- :441-468 ----------------------------------------------------------------------------------
def runtime_if_nested_while(infield: F64, outfield: F64):
    with computation(PARALLEL), interval(...):
        if infield < 10:
            outfield = 1
            done = False
            while not done:
                outfield = 2
                done = True
        else:
            condition = True
            while condition:
                outfield = 4
                condition = False
            outfield = 3


# ---- :471-489 ----------------------------------------------------------------------------------
def ternary_op(infield: F64, outfield: F64):
    with computation(PARALLEL), interval(...):
        outfield = infield if infield > 0.0 else -infield[0, 1, 0]


def _ternary_expected(a, p, ext, domain):
    x = a["infield"]
    return {"outfield": (x[:, :-1] > 0.0) * x[:, :-1] + (x[:, :-1] <= 0.0) * (-x[:, 1:])}


# ---- :492-533 ----------------------------------------------------------------------------------
def three_way_and(outfield: F64, *, a: float, b: float, c: float):
    with computation(PARALLEL), interval(...):
        if a > 0 and b > 0 and c > 0:
            outfield = 1
        else:
            outfield = 0


def three_way_or(outfield: F64, *, a: float, b: float, c: float):
    with computation(PARALLEL), interval(...):
        if a > 0 or b > 0 or c > 0:
            outfield = 1
        else:
            outfield = 0


# ---- :536-561 with stencil_definitions.py:406-421 ------------------------------------------------
def optional_field(in_field: F64, out_field: F64, dyn_tend: F64, phys_tend: F64 = None, *, dt: float):
    from __externals__ import PHYS_TEND

    with computation(PARALLEL), interval(...):
        out_field = in_field + dt * dyn_tend
        if __INLINED(PHYS_TEND):
            out_field = out_field + dt * phys_tend


def _optional_expected(a, p, ext, domain):
    out = a["in_field"] + p["dt"] * a["dyn_tend"]
    if ext["PHYS_TEND"]:
        out = out + p["dt"] * a["phys_tend"]
    return {"out_field": out}


# ---- :700-762 ----------------------------------------------------------------------------------
def read_outside_k_interval_1(field_in: F64, field_out: F64):
    with computation(PARALLEL), interval(...):
        field_out = field_in[0, 0, -1] + field_in[0, 0, 1]


def read_outside_k_interval_2(field_in: F64, field_out: F64):
    with computation(PARALLEL), interval(-1, None):
        field_out = field_in[0, 0, 1]


def read_outside_k_interval_3(field_in: F64, field_out: F64):
    with computation(PARALLEL), interval(0, 1):
        field_out = field_in[0, 0, -1]


def _rok2_expected(a, p, ext, domain):
    out = a["field_out"].copy()
    out[:, :, -1] = a["field_in"][:, :, domain[2]]
    return {"field_out": out}


def _rok3_expected(a, p, ext, domain):
    out = a["field_out"].copy()
    out[:, :, 0] = a["field_in"][:, :, 0]
    return {"field_out": out}


# ---- :811-832 ----------------------------------------------------------------------------------
def diagonal_k_offset(field_in: F64, field_out: F64):
    with computation(PARALLEL), interval(...):
        field_out = field_in[0, 0, 1]
    with computation(PARALLEL), interval(0, -1):
        field_out += field_in[0, -1, 1]


def _diagonal_expected(a, p, ext, domain):
    x = a["field_in"]
    out = x[:, 1:, 1:].copy()
    out[:, :, :-1] += x[:, :-1, 1:-1]
    return {"field_out": out}


# ---- :835-943 (horizontal regions) ---------------------------------------------------------------
def horizontal_regions(field_in: F32, field_out: F32):
    with computation(PARALLEL), interval(...):
        field_out = field_in
        with horizontal(region[I[0], :], region[I[-1], :]):
            field_out = field_in + 1.0
        with horizontal(region[:, J[0]], region[:, J[-1]]):
            field_out = field_in - 1.0


def _regions_expected(a, p, ext, domain, start=None):
    x = a["field_in"]
    out = x.copy() if start is None else np.full_like(x, start)
    out[0] = x[0] + 1.0
    out[-1] = x[-1] + 1.0
    out[:, 0] = x[:, 0] - 1.0
    out[:, -1] = x[:, -1] - 1.0
    return {"field_out": out}


def horizontal_regions_partial_writes(field_in: F32, field_out: F32):
    with computation(PARALLEL), interval(...):
        with horizontal(region[I[0], :], region[I[-1], :]):
            field_out = field_in + 1.0
        with horizontal(region[:, J[0]], region[:, J[-1]]):
            field_out = field_in - 1.0


def horizontal_regions_corners(field_in: F32, field_out: F32):
    with computation(PARALLEL), interval(...):
        with horizontal(region[I[0] : I[0] + 2, J[0] : J[0] + 2], region[I[-1] - 2 : I[-1], J[-1] - 2 : J[-1]]):
            field_out = field_in + 1.0
        with horizontal(region[I[0] : I[0] + 2, J[-1] - 2 : J[-1]], region[I[-1] - 2 : I[-1], J[0] : J[0] + 2]):
            field_out = field_in - 1.0


def _corners_expected(a, p, ext, domain):
    x = a["field_in"]
    out = np.full_like(x, 42)
    out[0:2, 0:2] = x[0:2, 0:2] + 1.0
    out[-3:-1, -3:-1] = x[-3:-1, -3:-1] + 1.0
    out[0:2, -3:-1] = x[0:2, -3:-1] - 1.0
    out[-3:-1, 0:2] = x[-3:-1, 0:2] - 1.0
    return {"field_out": out}


# ---- :765-808 (run-time K index) -------------------------------------------------------------------
def variable_k_read(field_in: F32, field_out: F32, index: "Field[K, np.int32]"):
    with computation(PARALLEL), interval(1, None):
        field_out = field_in[0, 0, index]


def _variable_k_expected(a, p, ext, domain):
    x, index = a["field_in"], a["index"]
    out = a["field_out"].copy()
    out[:, :, 1:] = x[:, :, (np.arange(x.shape[-1]) + index)[1:]]
    return {"field_out": out}


def variable_k_and_read_outside(field_in: F64, field_out: F64, index: "Field[K, np.int32]"):
    with computation(PARALLEL), interval(1, None):
        field_out[0, 0, 0] = field_in[0, 0, index] + field_in[0, 0, -2]


def _variable_k_outside_expected(a, p, ext, domain):
    x, index = a["field_in"], a["index"]  # field_in carries one extra level below the domain
    out = a["field_out"].copy()
    idx = 1 + (np.arange(domain[-1]) + index)[1:]
    out[:, :, 1:] = x[:, :, idx]
    out[:, :, 1:] += x[:, :, :-2]
    return {"field_out": out}


# ---- :614-697 (lower-dimensional fields with data dimensions) ----------------------------------------
def non_3d_fields(field_in: "Field[K, np.float64]", another_field: "Field[IJ, (np.float64, (3, 2, 2))]",
                  field_out: "Field[(np.float64, (3, 2))]"):
    with computation(PARALLEL), interval(...):
        field_out[0, 0, 0][0, 0] = field_in[0] + another_field[-1, -1][0, 0, 0] + another_field[-1, -1][0, 0, 1]
        field_out[0, 0, 0][0, 1] = 2 * (
            another_field[-1, -1][1, 0, 0]
            + another_field[-1, -1][1, 0, 1]
            + another_field[-1, -1][1, 1, 0]
            + another_field[-1, -1][1, 1, 1]
        )

        field_out[0, 0, 0][1, 0] = field_in[0] + another_field[1, 1][0, 0, 0] + another_field[1, 1][0, 0, 1]
        field_out[0, 0, 0][1, 1] = 3 * (
            another_field[1, 1][1, 0, 0]
            + another_field[1, 1][1, 0, 1]
            + another_field[1, 1][1, 1, 0]
            + another_field[1, 1][1, 1, 1]
        )

        field_out[0, 0, 0][2, 0] = field_in[0] + another_field[0, 0][0, 0, 0] + another_field[-1, 1][0, 0, 1]
        field_out[0, 0, 0][2, 1] = 4 * (
            another_field[-1, 1][1, 0, 0]
            + another_field[-1, 1][1, 0, 1]
            + another_field[-1, 1][1, 1, 0]
            + another_field[-1, 1][1, 1, 1]
        )


def _non_3d_expected(a, p, ext, domain):
    fi, af = a["field_in"], a["another_field"]
    out = a["field_out"].copy()
    out[:, :, :, 0, 0] = fi[:] + af[:-2, :-2, None, 0, 0, 0] + af[:-2, :-2, None, 0, 0, 1]
    out[:, :, :, 0, 1] = 2 * (af[:-2, :-2, None, 1, 0, 0] + af[:-2, :-2, None, 1, 0, 1] + af[:-2, :-2, None, 1, 1, 0]
                              + af[:-2, :-2, None, 1, 1, 1])
    out[:, :, :, 1, 0] = fi[:] + af[2:, 2:, None, 0, 0, 0] + af[2:, 2:, None, 0, 0, 1]
    out[:, :, :, 1, 1] = 3 * (af[2:, 2:, None, 1, 0, 0] + af[2:, 2:, None, 1, 0, 1] + af[2:, 2:, None, 1, 1, 0]
                              + af[2:, 2:, None, 1, 1, 1])
    out[:, :, :, 2, 0] = fi[:] + af[1:-1, 1:-1, None, 0, 0, 0] + af[:-2, 2:, None, 0, 0, 1]
    out[:, :, :, 2, 1] = 4 * (af[:-2, 2:, None, 1, 0, 0] + af[:-2, 2:, None, 1, 0, 1] + af[:-2, 2:, None, 1, 1, 0]
                              + af[:-2, 2:, None, 1, 1, 1])
    return {"field_out": out}


# ---- :946-977 ----------------------------------------------------------------------------------
def typed_temporary(field_in: F32, field_out: F32):
    tmp: Field[(np.float32, (2, 2))] = 0
    with computation(PARALLEL):
        with interval(0, -1):
            tmp[0, 0, 0][0, 0] = field_in[0, 0, 0]
            tmp[0, 0, 0][1, 0] = field_in[0, 0, 1]
            tmp[0, 0, 0][0, 1] = -1.0
            tmp[0, 0, 0][1, 1] = -1.0
            field_out = tmp[0, 0, 0][0, 0] + tmp[0, 0, 0][1, 0]
        with interval(-1, None):
            field_out = 0


def _typed_temporary_expected(a, p, ext, domain):
    x = a["field_in"]
    out = a["field_out"].copy()
    out[:, :, :-1] = x[:, :, :-1] + x[:, :, 1:]
    out[:, :, -1] = 0
    return {"field_out": out}


# ---- :980-1060 (vector-valued statements over data dimensions) ----------------------------------------
def vector_gen_assignment(field_in: "Field[(np.float64, (2,))]", field_out: "Field[(np.float64, (2,))]"):
    with computation(PARALLEL), interval(...):
        field_out = 2 * field_in


def matrix_assignment(field_in: "Field[(np.float32, (2, 3))]", field_out: "Field[(np.float32, (2, 3))]"):
    with computation(PARALLEL), interval(...):
        field_out = field_in


def vector_vector_op(field_1: "Field[(np.float32, (2,))]", field_2: "Field[(np.float32, (2,))]",
                     field_out: "Field[(np.float32, (2,))]"):
    with computation(PARALLEL), interval(...):
        field_out = field_1 + field_2


# ---- :1045-1092 ---------------------------------------------------------------------------------
def combined_vector_scalar_op(field_1: "Field[(np.float64, (2,))]", field_2: "Field[(np.float64, (2,))]",
                              field_out: "Field[(np.float64, (2,))]"):
    with computation(PARALLEL), interval(...):
        field_out = 3 * (field_1 + field_2 * field_2)


def vectorized_temporary(field_in: "Field[(np.float32, (2,))]", field_out: "Field[(np.float32, (2,))]"):
    tmp: Field[(np.float32, (2,))] = 0
    with computation(PARALLEL), interval(...):
        tmp[0, 0, 0][0] = 2
        tmp[0, 0, 0][1] = 3
        field_out = tmp * field_in


def _vectorized_temporary_expected(a, p, ext, domain):
    x = a["field_in"]
    out = a["field_out"].copy()
    out[:, :, :, 0] = 2 * x[:, :, :, 0]
    out[:, :, :, 1] = 3 * x[:, :, :, 1]
    return {"field_out": out}


# ---- :1097-1162 (matrix @ vector, matrix.T @ vector) -------------------------------------------------
def matmul(matrix: "Field[(np.float64, (4, 6))]", field_1: "Field[(np.float64, (6,))]", field_2: "Field[(np.float64, (4,))]"):
    with computation(PARALLEL):
        with interval(0, 1):
            field_2 = matrix @ field_1
        with interval(1, 2):
            field_1 = matrix.T @ field_2


def masked_matmul(matrix: "Field[K, (np.float64, (4, 6))]", field_1: "Field[(np.float64, (6,))]",
                  field_2: "Field[(np.float64, (4,))]"):
    with computation(PARALLEL):
        with interval(0, 1):
            field_2 = matrix @ field_1
        with interval(1, 2):
            field_1 = matrix.T @ field_2


def _ordered_dot(m, v):
    """sum_c m[..., r, c] * v[..., c], accumulated left to right from the first product -- the order the
    reference's unrolling of '@' fixes (defir_to_gtir.py:265-273); einsum may associate differently."""
    acc = m[..., :, 0] * v[..., None, 0]
    for c in range(1, m.shape[-1]):
        acc = acc + m[..., :, c] * v[..., None, c]
    return acc


def _matmul_expected(a, p, ext, domain):
    m, f1, f2 = a["matrix"], a["field_1"].copy(), a["field_2"].copy()
    m0, m1 = (m[0], m[1]) if m.ndim == 3 else (m[:, :, 0], m[:, :, 1])
    f2[:, :, 0] = _ordered_dot(m0, f1[:, :, 0])
    f1[:, :, 1] = _ordered_dot(np.swapaxes(m1, -1, -2), f2[:, :, 1])
    # the reference's own validation (einsum), to rounding
    sub = "lm" if m.ndim == 3 else "ijlm"
    np.testing.assert_allclose(f2[:, :, 0], np.einsum(sub + ",ijm->ijl", m0, a["field_1"][:, :, 0]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(f1[:, :, 1], np.einsum(sub + ",ijl->ijm", m1, a["field_2"][:, :, 1]), rtol=1e-12, atol=1e-12)
    return {"field_1": f1, "field_2": f2}


# ---- :564-611 with stencil_definitions.py:424-447 --------------------------------------------------
def two_optional_fields(in_a: F64, in_b: F64, out_a: F64, out_b: F64, dyn_tend_a: F64, dyn_tend_b: F64,
                        phys_tend_a: F64 = None, phys_tend_b: F64 = None, *, dt: float):
    from __externals__ import PHYS_TEND_A, PHYS_TEND_B

    with computation(PARALLEL), interval(...):
        out_a = in_a + dt * dyn_tend_a
        out_b = in_b + dt * dyn_tend_b
        if __INLINED(PHYS_TEND_A):
            out_a = out_a + dt * phys_tend_a
        if __INLINED(PHYS_TEND_B):
            out_b = out_b + dt * phys_tend_b


def _two_optional_expected(a, p, ext, domain):
    out_a = a["in_a"] + p["dt"] * a["dyn_tend_a"]
    out_b = a["in_b"] + p["dt"] * a["dyn_tend_b"]
    if ext["PHYS_TEND_A"]:
        out_a = out_a + p["dt"] * a["phys_tend_a"]
    if ext["PHYS_TEND_B"]:
        out_b = out_b + p["dt"] * a["phys_tend_b"]
    return {"out_a": out_a, "out_b": out_b}


R10 = (-10.0, 10.0)
R1 = (-1.0, 1.0)
SUITES: Dict[str, Suite] = {
    "identity": Suite(identity, {"field_a": (np.float64, Z, R10)}, lambda a, p, e, d: {"field_a": a["field_a"]}),
    "copy": Suite(copy, {"field_a": (np.float64, Z, R10), "field_b": (np.float64, Z, R10)},
                  lambda a, p, e, d: {"field_b": a["field_a"]}),
    "aug_assign": Suite(aug_assign, {"field_a": (np.float64, Z, R10), "field_b": (np.float64, Z, R10)},
                        lambda a, p, e, d: {"field_a": (a["field_a"] + 1.0) * 2.0, "field_b": (a["field_b"] - 1.0) / 2.0}),
    "global_scale": Suite(global_scale, {"field_a": (np.float64, Z, R1)},
                          lambda a, p, e, d: {"field_a": e["SCALE_FACTOR"] * a["field_a"]},
                          externals=({"SCALE_FACTOR": 1.0}, {"SCALE_FACTOR": 1e3}, {"SCALE_FACTOR": 1e6})),
    "parametric_scale": Suite(parametric_scale, {"field_a": (np.float64, Z, R10)},
                              lambda a, p, e, d: {"field_a": p["scale"] * a["field_a"]}, params={"scale": (-100, 100)}),
    "parametric_mix": Suite(parametric_mix,
                            {"field_a": (np.float64, Z, R10), "field_b": (np.float64, Z, R10),
                             "field_c": (np.float64, Z, R10), "field_out": (np.float32, Z, R10)},
                            _parametric_mix_expected, params={"weight": (-10, 10), "alpha_factor": (-1, 1)},
                            externals=({"USE_ALPHA": True}, {"USE_ALPHA": False})),
    "heat_equation_ftcs": Suite(heat_equation_ftcs,
                                {"u": (np.float64, ((1, 1), (0, 0), (0, 0)), R10), "v": (np.float64, ((0, 0), (1, 1), (0, 0)), R10),
                                 "u_new": (np.float64, Z, R10), "v_new": (np.float64, Z, R10)},
                                _heat_expected, params={"ru": (0, 0.5), "rv": (0, 0.5)}),
    "horizontal_diffusion": Suite(horizontal_diffusion,
                                  {"u": (np.float64, ((2, 2), (2, 2), (0, 0)), R10), "diffusion": (np.float64, Z, R10)},
                                  _hdiff_expected, params={"weight": (0, 0.5)}),
    "horizontal_diffusion_subroutines": Suite(horizontal_diffusion_subroutines,
                                              {"u": (np.float64, ((2, 2), (2, 2), (0, 0)), R10), "diffusion": (np.float64, Z, R10)},
                                              _hdiff_expected, params={"weight": (0, 0.5)},
                                              externals=({"fwd_diff": wrap1arg2return},)),
    "horizontal_diffusion_subroutines2": Suite(horizontal_diffusion_subroutines2,
                                               {"u": (np.float64, ((2, 2), (2, 2), (0, 0)), R10), "diffusion": (np.float64, Z, R10)},
                                               _hdiff_expected, params={"weight": (0, 0.5)},
                                               externals=({"BRANCH": True}, {"BRANCH": False})),
    "runtime_if_flat": Suite(runtime_if_flat, {"outfield": (np.float64, Z, R10)},
                             lambda a, p, e, d: {"outfield": np.full_like(a["outfield"], 1)}),
    "runtime_if_nested": Suite(runtime_if_nested, {"outfield": (np.float64, Z, R10)},
                               lambda a, p, e, d: {"outfield": np.full_like(a["outfield"], 2)}),
    "three_fold_nested_if": Suite(three_fold_nested_if, {"field_a": (np.float64, Z, R1)}, _three_fold_expected,
                                  domains=((3, 3, 3),)),
    "runtime_if_nested_data_dependent": Suite(runtime_if_nested_data_dependent,
                                              {"field_a": (np.float64, Z, R1), "field_b": (np.float64, Z, R1), "field_c": (np.float64, Z, R1)},
                                              _data_dependent_expected, params={"factor": (-100, 100)}, domains=((3, 3, 3), (5, 4, 3))),
    "runtime_if_nested_while": Suite(runtime_if_nested_while,
                                     {"infield": (np.float64, Z, R1), "outfield": (np.float64, Z, R10)},
                                     lambda a, p, e, d: {"outfield": np.full_like(a["outfield"], 2)}),
    "ternary_op": Suite(ternary_op, {"infield": (np.float64, ((0, 0), (0, 1), (0, 0)), R10), "outfield": (np.float64, Z, R10)},
                        _ternary_expected, domains=((1, 2, 1), (3, 4, 5), (15, 14, 13))),
    "three_way_and": Suite(three_way_and, {"outfield": (np.float64, Z, R10)},
                           lambda a, p, e, d: {"outfield": np.full_like(a["outfield"], 1 if p["a"] > 0 and p["b"] > 0 and p["c"] > 0 else 0)},
                           params={"a": (-100, 100), "b": (-100, 100), "c": (-100, 100)}),
    "three_way_or": Suite(three_way_or, {"outfield": (np.float64, Z, R10)},
                          lambda a, p, e, d: {"outfield": np.full_like(a["outfield"], 1 if p["a"] > 0 or p["b"] > 0 or p["c"] > 0 else 0)},
                          params={"a": (-100, 100), "b": (-100, 100), "c": (-100, 100)}),
    "optional_field": Suite(optional_field,
                            {"in_field": (np.float64, Z, R10), "out_field": (np.float64, Z, R10),
                             "dyn_tend": (np.float64, Z, R10), "phys_tend": (np.float64, Z, R10)},
                            _optional_expected, params={"dt": (0, 100)},
                            externals=({"PHYS_TEND": False}, {"PHYS_TEND": True}), optional={"phys_tend": "PHYS_TEND"}),
    "read_outside_k_interval_1": Suite(read_outside_k_interval_1,
                                       {"field_in": (np.float64, ((0, 0), (0, 0), (1, 1)), R10), "field_out": (np.float64, Z, R10)},
                                       lambda a, p, e, d: {"field_out": a["field_in"][:, :, 0:-2] + a["field_in"][:, :, 2:]},
                                       domains=((4, 4, 4), (3, 2, 1))),
    "read_outside_k_interval_2": Suite(read_outside_k_interval_2,
                                       {"field_in": (np.float64, ((0, 0), (0, 0), (0, 1)), R10), "field_out": (np.float64, Z, R10)},
                                       _rok2_expected, domains=((4, 4, 4),)),
    "read_outside_k_interval_3": Suite(read_outside_k_interval_3,
                                       {"field_in": (np.float64, ((0, 0), (0, 0), (1, 0)), R10), "field_out": (np.float64, Z, R10)},
                                       _rok3_expected, domains=((4, 4, 4),)),
    "diagonal_k_offset": Suite(diagonal_k_offset,
                               {"field_in": (np.float64, ((0, 0), (1, 0), (0, 1)), (0.1, 10.0)), "field_out": (np.float64, Z, (0.1, 10.0))},
                               _diagonal_expected, domains=((2, 2, 2), (2, 2, 8), (5, 6, 7))),
    "variable_k_read": Suite(variable_k_read,
                             {"field_in": (np.float32, Z, R10), "field_out": (np.float32, Z, R10), "index": (np.int32, Z, (-1, 0))},
                             _variable_k_expected, domains=((2, 2, 2), (2, 2, 8), (5, 4, 9)), axes={"index": "K"}),
    "variable_k_and_read_outside": Suite(variable_k_and_read_outside,
                                         {"field_in": (np.float64, ((0, 0), (0, 0), (1, 0)), (0.1, 10.0)),
                                          "field_out": (np.float64, Z, (0.1, 10.0)), "index": (np.int32, Z, (-1, 0))},
                                         _variable_k_outside_expected, domains=((2, 2, 2), (2, 2, 8), (5, 4, 9)),
                                         axes={"index": "K"}),
    "non_3d_fields": Suite(non_3d_fields,
                           {"field_in": (np.float64, Z, R10), "another_field": (np.float64, ((1, 1), (1, 1), (0, 0)), R10),
                            "field_out": (np.float64, Z, R10)},
                           _non_3d_expected, domains=((4, 4, 4), (10, 7, 5)), axes={"field_in": "K", "another_field": "IJ"},
                           data_dims={"another_field": (3, 2, 2), "field_out": (3, 2)}),
    "typed_temporary": Suite(typed_temporary, {"field_in": (np.float32, Z, R10), "field_out": (np.float32, Z, R10)},
                             _typed_temporary_expected, domains=((2, 2, 2), (2, 2, 8), (5, 3, 6))),
    "vector_gen_assignment": Suite(vector_gen_assignment, {"field_in": (np.float64, Z, R10), "field_out": (np.float64, Z, R10)},
                                   lambda a, p, e, d: {"field_out": 2 * a["field_in"]}, domains=((2, 2, 2), (5, 4, 3)),
                                   data_dims={"field_in": (2,), "field_out": (2,)}),
    "matrix_assignment": Suite(matrix_assignment, {"field_in": (np.float32, Z, R10), "field_out": (np.float32, Z, R10)},
                               lambda a, p, e, d: {"field_out": a["field_in"]}, domains=((2, 2, 2), (5, 4, 3)),
                               data_dims={"field_in": (2, 3), "field_out": (2, 3)}),
    "vector_vector_op": Suite(vector_vector_op,
                              {"field_1": (np.float32, Z, R10), "field_2": (np.float32, Z, R10), "field_out": (np.float32, Z, R10)},
                              lambda a, p, e, d: {"field_out": a["field_1"] + a["field_2"]}, domains=((2, 2, 2), (5, 4, 3)),
                              data_dims={"field_1": (2,), "field_2": (2,), "field_out": (2,)}),
    "combined_vector_scalar_op": Suite(combined_vector_scalar_op,
                                       {"field_1": (np.float64, Z, (1.0, 10.0)), "field_2": (np.float64, Z, (1.0, 10.0)),
                                        "field_out": (np.float64, Z, (1.0, 10.0))},
                                       lambda a, p, e, d: {"field_out": 3 * (a["field_1"] + a["field_2"] * a["field_2"])},
                                       domains=((2, 2, 2), (5, 4, 3)),
                                       data_dims={"field_1": (2,), "field_2": (2,), "field_out": (2,)}),
    "vectorized_temporary": Suite(vectorized_temporary, {"field_in": (np.float32, Z, R10), "field_out": (np.float32, Z, R10)},
                                  _vectorized_temporary_expected, domains=((2, 2, 2), (5, 4, 3)),
                                  data_dims={"field_in": (2,), "field_out": (2,)}),
    "matmul": Suite(matmul, {"matrix": (np.float64, Z, R10), "field_1": (np.float64, Z, R10), "field_2": (np.float64, Z, R10)},
                    _matmul_expected, domains=((2, 2, 2), (5, 4, 3)),
                    data_dims={"matrix": (4, 6), "field_1": (6,), "field_2": (4,)}),
    "masked_matmul": Suite(masked_matmul,
                           {"matrix": (np.float64, Z, R10), "field_1": (np.float64, Z, R10), "field_2": (np.float64, Z, R10)},
                           _matmul_expected, domains=((2, 2, 2), (5, 4, 3)), axes={"matrix": "K"},
                           data_dims={"matrix": (4, 6), "field_1": (6,), "field_2": (4,)}),
    "two_optional_fields": Suite(two_optional_fields,
                                 {n: (np.float64, Z, R10) for n in ("in_a", "in_b", "out_a", "out_b", "dyn_tend_a", "dyn_tend_b",
                                                                    "phys_tend_a", "phys_tend_b")},
                                 _two_optional_expected, params={"dt": (0, 100)},
                                 externals=({"PHYS_TEND_A": False, "PHYS_TEND_B": False}, {"PHYS_TEND_A": False, "PHYS_TEND_B": True},
                                            {"PHYS_TEND_A": True, "PHYS_TEND_B": True}),
                                 optional={"phys_tend_a": "PHYS_TEND_A", "phys_tend_b": "PHYS_TEND_B"},
                                 domains=((1, 1, 1), (7, 5, 3))),
    "horizontal_regions": Suite(horizontal_regions, {"field_in": (np.float32, Z, R10), "field_out": (np.float32, Z, R10)},
                                _regions_expected, domains=((4, 4, 2), (9, 7, 3))),
    "horizontal_regions_partial_writes": Suite(horizontal_regions_partial_writes,
                                               {"field_in": (np.float32, Z, R10), "field_out": (np.float32, Z, (42.0, 42.0))},
                                               lambda a, p, e, d: _regions_expected(a, p, e, d, start=42),
                                               domains=((4, 4, 2), (9, 7, 3))),
    "horizontal_regions_corners": Suite(horizontal_regions_corners,
                                        {"field_in": (np.float32, Z, R10), "field_out": (np.float32, Z, (42.0, 42.0))},
                                        _corners_expected, domains=((4, 4, 2), (8, 9, 3))),
}


def cases():
    """(suite name, externals, domain) triples."""
    for name, suite in SUITES.items():
        for ext in suite.externals:
            for domain in suite.domains:
                yield name, ext, domain


def make_case(name: str, ext: Dict[str, Any], domain, seed: int = 1337):
    """Seeded inputs for one case -> (arrays incl. boundary, origins, params, expected outputs over the domain)."""
    suite = SUITES[name]
    rng = np.random.default_rng(seed)
    arrays, origins = {}, {}
    for fname, (dt, boundary, (lo, hi)) in suite.fields.items():
        present = [ax for ax, name in enumerate("IJK") if name in suite.axes.get(fname, "IJK")]
        shape = tuple(domain[ax] + boundary[ax][0] + boundary[ax][1] for ax in present) + tuple(suite.data_dims.get(fname, ()))
        if np.dtype(dt).kind in "iu":
            arrays[fname] = rng.integers(int(lo), int(hi) + 1, shape).astype(dt)
        else:
            arrays[fname] = rng.uniform(lo, hi, shape).astype(dt)
        origins[fname] = tuple(boundary[ax][0] for ax in present) + (0,) * len(suite.data_dims.get(fname, ()))
    params = {p: float(rng.uniform(lo, hi)) for p, (lo, hi) in suite.params.items()}
    expected = suite.expected({k: v.copy() for k, v in arrays.items()}, params, ext, domain)
    return arrays, origins, params, expected
