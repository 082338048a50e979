"""Small GTScript programs for the planner's rewriting passes (tests/test_codegen.py): in a file of their own because the
frontend reads a definition's source."""
import numpy as np

from gt4py_amd.cartesian.gtscript import BACKWARD, FORWARD, PARALLEL, Field, computation, interval  # noqa: F401

F64 = Field[np.float64]


def boundary_and_interior(a: F64, out: F64):
    """`lap` and `flux` are used by both interval blocks, each for itself: one temporary per block."""
    with computation(PARALLEL):
        with interval(0, 1):
            lap = a[1, 0, 0] + a[-1, 0, 0] - 2.0 * a
            flux = lap[1, 0, 0] - lap
            out = a + flux
        with interval(1, None):
            lap = a[1, 0, 0] + a[-1, 0, 0] + a[0, 1, 0] + a[0, -1, 0] - 4.0 * a
            flux = lap[1, 0, 0] - lap
            out = a + 0.5 * flux


def value_crosses_blocks(a: F64, out: F64):
    """`t` flows from the first computation into the second: it stays one temporary (in memory)."""
    with computation(PARALLEL), interval(...):
        t = a * 2.0
    with computation(PARALLEL), interval(...):
        out = t[1, 0, 0] + t


def if_else_pair(a: F64, b: F64, out: F64, other: F64):
    """The two branches assign `out`: one conditional expression.  `other` is only assigned in one branch: it keeps its
    old value elsewhere and stays a conditional assignment."""
    with computation(PARALLEL), interval(...):
        if a > 0.0:
            out = a + b[1, 0, 0]
            other = b
        else:
            out = a - b[-1, 0, 0]


def if_else_with_interference(a: F64, b: F64, out: F64):
    """The `else` branch reads what the `if` branch assigned in between (`t`): the pair must NOT be merged into one
    expression evaluated at the position of the second assignment."""
    with computation(PARALLEL), interval(...):
        t = b
        if a > 0.0:
            out = t
            t = a * 3.0
        else:
            out = t + 1.0
            t = a


def boundary_only_write_read_back(inp: F64, acc: F64, s: F64, out: F64):
    """`acc` is assigned at the first level only and read back by the second sweep on ALL levels: above the first level
    the second sweep must see the caller's `acc`, so it cannot live in the on-chip top-of-column cache (only `s`, which
    every level of the first sweep assigns, can)."""
    with computation(FORWARD):
        with interval(0, 1):
            acc = inp
            s = inp
        with interval(1, None):
            s = s[0, 0, -1] + inp
    with computation(BACKWARD):
        with interval(-1, None):
            out = acc + s
        with interval(0, -1):
            out = out[0, 0, 1] + acc * s


def condition_input_rewritten_between_branches(a: F64, f: F64, t0: F64, out: F64):
    """`not (f > 0)` is evaluated AFTER `f` changed sign: the two conditional assignments of `t` are not the branches of one
    `if` / `else`, where neither holds `t` keeps the value of the first statement."""
    with computation(PARALLEL), interval(...):
        t = t0
        if f > 0.0:
            t = a
        f = -f
        if not (f > 0.0):
            t = a * 2.0
        out = t
