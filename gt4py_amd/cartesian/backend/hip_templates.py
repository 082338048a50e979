"""The stencil shapes the hand-written gfx950 kernels implement, written as GTScript.

Each function below is the definition a kernel family reproduces bit-for-bit; the ``hip:mi300``
backend parses them with the same frontend as user code and accepts a user stencil when its typed
IR is identical up to renaming of fields, parameters and temporaries (alpha-equivalence).  Matching
on the *parsed tree* rather than on source text means parenthesisation, whitespace, variable names,
argument order and the ``f[I + 1]`` / ``f[1, 0, 0]`` spellings do not matter -- but a different
association of floating-point operations (which changes results) does, and is rejected.

Sources of the shapes (reference file:line):
  lap_notebook            /root/reference/examples/lap_cartesian_vs_next.ipynb cell 7
  lap_docs                /root/reference/docs/user/cartesian/index.rst:24-28
  lap_suite               /root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/test_suites.py:214
  lap_avg                 /root/reference/tests/cartesian_tests/integration_tests/feature_tests/test_call_interface.py:159-164
  hdiff_limiter_field     .../multi_feature_tests/stencil_definitions.py:316-328 (= examples/cartesian/demo_horizontal_diffusion.ipynb cell 7)
  hdiff_plain_field       .../multi_feature_tests/stencil_definitions.py:206-216
  hdiff_plain_scalar      .../multi_feature_tests/test_suites.py:212-220 (weight is a scalar parameter)
  hdiff_limiter_scalar    the limiter form with a scalar coefficient (combination of the two above)
  tridiagonal_solver      .../multi_feature_tests/stencil_definitions.py:219-232

Dtype placeholders: ``"T"`` = field dtype, ``"S"`` = scalar parameter dtype.
"""

# The names below are only parsed, never executed.
from ..gtscript import BACKWARD, FORWARD, PARALLEL, Field, I, J, computation, interval  # noqa: F401


def lap_notebook(inp: Field["T"], out: Field["T"]):  # noqa: F821
    with computation(PARALLEL), interval(...):
        out = -4.0 * inp[0, 0, 0] + inp[-1, 0, 0] + inp[1, 0, 0] + inp[0, -1, 0] + inp[0, 1, 0]  # noqa: F841


def lap_docs(inp: Field["T"], out: Field["T"]):  # noqa: F821
    with computation(PARALLEL), interval(...):
        out = -4.0 * inp + (inp[I + 1] + inp[I - 1] + inp[J + 1] + inp[J - 1])  # noqa: F841


def lap_suite(inp: Field["T"], out: Field["T"]):  # noqa: F821
    with computation(PARALLEL), interval(...):
        out = 4.0 * inp[0, 0, 0] - (inp[1, 0, 0] + inp[-1, 0, 0] + inp[0, 1, 0] + inp[0, -1, 0])  # noqa: F841


def lap_avg(inp: Field["T"], out: Field["T"]):  # noqa: F821
    with computation(PARALLEL), interval(...):
        out = 0.25 * (+inp[0, 1, 0] + inp[0, -1, 0] + inp[1, 0, 0] + inp[-1, 0, 0])  # noqa: F841


def hdiff_limiter_field(in_field: Field["T"], out_field: Field["T"], coeff: Field["T"]):  # noqa: F821
    with computation(PARALLEL), interval(...):
        lap_field = 4.0 * in_field[0, 0, 0] - (
            in_field[1, 0, 0] + in_field[-1, 0, 0] + in_field[0, 1, 0] + in_field[0, -1, 0]
        )
        res = lap_field[1, 0, 0] - lap_field[0, 0, 0]
        flx_field = 0 if (res * (in_field[1, 0, 0] - in_field[0, 0, 0])) > 0 else res
        res = lap_field[0, 1, 0] - lap_field[0, 0, 0]
        fly_field = 0 if (res * (in_field[0, 1, 0] - in_field[0, 0, 0])) > 0 else res
        out_field = in_field[0, 0, 0] - coeff[0, 0, 0] * (  # noqa: F841
            flx_field[0, 0, 0] - flx_field[-1, 0, 0] + fly_field[0, 0, 0] - fly_field[0, -1, 0]
        )


def hdiff_limiter_scalar(in_field: Field["T"], out_field: Field["T"], *, coeff: "S"):  # noqa: F821
    with computation(PARALLEL), interval(...):
        lap_field = 4.0 * in_field[0, 0, 0] - (
            in_field[1, 0, 0] + in_field[-1, 0, 0] + in_field[0, 1, 0] + in_field[0, -1, 0]
        )
        res = lap_field[1, 0, 0] - lap_field[0, 0, 0]
        flx_field = 0 if (res * (in_field[1, 0, 0] - in_field[0, 0, 0])) > 0 else res
        res = lap_field[0, 1, 0] - lap_field[0, 0, 0]
        fly_field = 0 if (res * (in_field[0, 1, 0] - in_field[0, 0, 0])) > 0 else res
        out_field = in_field[0, 0, 0] - coeff * (  # noqa: F841
            flx_field[0, 0, 0] - flx_field[-1, 0, 0] + fly_field[0, 0, 0] - fly_field[0, -1, 0]
        )


def hdiff_plain_field(in_field: Field["T"], coeff: Field["T"], out_field: Field["T"]):  # noqa: F821
    with computation(PARALLEL), interval(...):
        lap_field = 4.0 * in_field[0, 0, 0] - (
            in_field[1, 0, 0] + in_field[-1, 0, 0] + in_field[0, 1, 0] + in_field[0, -1, 0]
        )
        flx_field = lap_field[1, 0, 0] - lap_field[0, 0, 0]
        fly_field = lap_field[0, 1, 0] - lap_field[0, 0, 0]
        out_field = in_field[0, 0, 0] - coeff[0, 0, 0] * (  # noqa: F841
            flx_field[0, 0, 0] - flx_field[-1, 0, 0] + fly_field[0, 0, 0] - fly_field[0, -1, 0]
        )


def hdiff_plain_scalar(in_field: Field["T"], out_field: Field["T"], *, coeff: "S"):  # noqa: F821
    with computation(PARALLEL), interval(...):
        lap_field = 4.0 * in_field[0, 0, 0] - (
            in_field[1, 0, 0] + in_field[-1, 0, 0] + in_field[0, 1, 0] + in_field[0, -1, 0]
        )
        flx_field = lap_field[1, 0, 0] - lap_field[0, 0, 0]
        fly_field = lap_field[0, 1, 0] - lap_field[0, 0, 0]
        out_field = in_field[0, 0, 0] - coeff * (  # noqa: F841
            flx_field[0, 0, 0] - flx_field[-1, 0, 0] + fly_field[0, 0, 0] - fly_field[0, -1, 0]
        )


def tridiagonal_solver(inf: Field["T"], diag: Field["T"], sup: Field["T"], rhs: Field["T"], out: Field["T"]):  # noqa: F821
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
