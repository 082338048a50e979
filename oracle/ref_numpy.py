"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in plain numpy slicing, of what gt4py's ``numpy`` backend computes
for the three hot-path stencils.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under
``gt4py_amd/`` does.

The reference cannot be imported in this image (Python 3.10 < 3.12), so
every function here is hand-derived from the reference's code-generation rules and cites
them.  Pinning status:

* ``laplacian``            pinned by the reference notebook KAT (x^2+y^2 -> 4) and the
                           avg-of-ones KAT (test_call_interface.py:221-285).
* ``hdiff`` (no limiter)   pinned by ``hdiff_validation`` = the reference's own numpy
                           validation function (test_suites.py:222-230).
* ``hdiff`` (flux limiter) parity UNPINNED by the reference (only smoke-run there); pinned
                           here by restatement + invariants (plane -> identity, limiter
                           inactive -> equals the pinned no-limiter form).
* ``tridiag``              parity UNPINNED by the reference (smoke-run on ones); pinned here
                           by restatement + scipy.linalg.solve_banded + residual checks.

Conventions shared by all functions (reference: src/gt4py/cartesian/utils/field.py:15-74,
src/gt4py/cartesian/gtc/numpy/npir_codegen.py:331-367):
  * arrays are indexed [i, j, k] whatever their memory strides are;
  * ``origin`` is the per-field index of the first compute-domain point, ``domain`` the
    (dI, dJ, dK) extent; only ``[origin, origin+domain)`` of written fields changes
    (docs/user/cartesian/gtscript.rst:99);
  * expressions are evaluated with exactly the parenthesisation Python's parser gives the
    stencil definition (left-associative + and -), one rounding per operation, no FMA.
"""

from __future__ import annotations

import numpy as np


def _win(a, origin, domain, lo=(0, 0), hi=(0, 0), off=(0, 0, 0)):
    """View of ``a`` over the compute domain grown by (lo, hi) in I,J and shifted by ``off``.

    Mirrors the slice strings the numpy backend emits: ``f[i+o : I+o, j+o : J+o, k:K]`` with
    ``i, I = _di_ - lo_i, _dI_ + hi_i`` (npir_codegen.py:35-43, 298-318) plus the per-field
    origin shift of the ``Field`` shim (cartesian/utils/field.py:34-74).
    """
    oi, oj, ok = origin
    di, dj, dk = domain
    i0 = oi - lo[0] + off[0]
    j0 = oj - lo[1] + off[1]
    k0 = ok + off[2]
    i1 = oi + di + hi[0] + off[0]
    j1 = oj + dj + hi[1] + off[1]
    k1 = ok + dk + off[2]
    assert i0 >= 0 and j0 >= 0 and k0 >= 0, "oracle: read outside of the array (low side)"
    assert i1 <= a.shape[0] and j1 <= a.shape[1] and k1 <= a.shape[2], (
        "oracle: read outside of the array (high side)"
    )
    return a[i0:i1, j0:j1, k0:k1]


# --------------------------------------------------------------------------------------
# 5-point Laplacian family (single statement, PARALLEL, interval(...))
# --------------------------------------------------------------------------------------
def laplacian(inp, out, *, origin_inp=(1, 1, 0), origin_out=(1, 1, 0), domain=None,
              variant="notebook"):
    """5-point star stencils, one rounding per op, in the parse order of the definition.

    variant "notebook": examples/lap_cartesian_vs_next.ipynb cell 7
        out = -4.0*inp[0,0,0] + inp[-1,0,0] + inp[1,0,0] + inp[0,-1,0] + inp[0,1,0]
        (``-4.0`` is UnaryOp(neg, 4.0) * inp -- frontend/gtscript_frontend.py:1477-1487 --
        which has the same value as (-4.0)*inp.)
    variant "docs": docs/user/cartesian/index.rst:24-28
        B = -4.*A + (A[I+1] + A[I-1] + A[J+1] + A[J-1])
    variant "suite": the ``4.0*u - (...)`` form of test_suites.py:214 /
        stencil_definitions.py:209-211 (the Laplacian inside horizontal diffusion)
    variant "avg": test_call_interface.py:159-164
        out = 0.25 * (+in[0,1,0] + in[0,-1,0] + in[1,0,0] + in[-1,0,0])
    Generated code shape: npir_codegen.py:205-212 (``({l} {op} {r})``), SURVEY Appendix A.1.
    """
    if domain is None:
        domain = (inp.shape[0] - 2, inp.shape[1] - 2, inp.shape[2])
    dt = inp.dtype.type

    def g(di, dj):
        return _win(inp, origin_inp, domain, off=(di, dj, 0))

    with np.errstate(divide="ignore", over="ignore", under="ignore", invalid="ignore"):
        if variant == "notebook":
            r = ((((-(dt(4.0))) * g(0, 0)) + g(-1, 0)) + g(1, 0)) + g(0, -1)
            r = r + g(0, 1)
        elif variant == "docs":
            r = ((-(dt(4.0))) * g(0, 0)) + (((g(1, 0) + g(-1, 0)) + g(0, 1)) + g(0, -1))
        elif variant == "suite":
            r = (dt(4.0) * g(0, 0)) - (((g(1, 0) + g(-1, 0)) + g(0, 1)) + g(0, -1))
        elif variant == "avg":
            r = dt(0.25) * ((((+g(0, 1)) + g(0, -1)) + g(1, 0)) + g(-1, 0))
        else:
            raise ValueError(variant)
        _win(out, origin_out, domain)[...] = r.astype(out.dtype, copy=False)
    return out


# --------------------------------------------------------------------------------------
# Horizontal diffusion (stencil_definitions.py:206-216 and :316-328)
# --------------------------------------------------------------------------------------
def hdiff(in_field, out_field, coeff, *, origin_in=(2, 2, 0), origin_out=(2, 2, 0),
          origin_coeff=None, domain=None, limiter=True, literal_float_precision=64):
    """Horizontal diffusion, statement by statement over the extended blocks.

    Follows SURVEY Appendix A.2: each statement is executed over its own horizontal block
    extent (gtc/passes/oir_optimizations/utils.py:293-313) before the next one
    (docs/user/cartesian/quickstart.rst:268-273):

        lap  on I(-1,1) x J(-1,1)
        flx  on I(-1,0) x J(0,0)      fly on I(0,0) x J(-1,0)
        out  on the compute domain

    dtype rules (gtc/passes/gtir_upcaster.py:43-143, gtir_dtype_resolver.py:54-57,
    frontend/gtscript_frontend.py:1250-1259): float literals are float64 unless
    ``literal_float_precision=32``; each binary op computes in the wider operand dtype;
    temporaries take the dtype of their first right-hand side; the int literal ``0`` in the
    ternary is cast to the other branch's dtype; the final right-hand side is rounded once to
    the dtype of ``out_field``.  For float32 fields with the default precision this makes
    lap/res/flx/fly float64 while ``in[1,0,0] - in[0,0,0]`` is a float32 subtraction that is
    widened afterwards.

    ``coeff`` is either an array (the field form of stencil_definitions.py) or a scalar
    (the ``weight`` parameter form of test_suites.py:205-220).
    """
    if domain is None:
        domain = (in_field.shape[0] - 4, in_field.shape[1] - 4, in_field.shape[2])
    lit = np.float64 if literal_float_precision == 64 else np.float32
    fdt = in_field.dtype.type

    def gin(di, dj, lo=(0, 0), hi=(0, 0)):
        return _win(in_field, origin_in, domain, lo, hi, (di, dj, 0))

    with np.errstate(divide="ignore", over="ignore", under="ignore", invalid="ignore"):
        # lap over I(-1,1) x J(-1,1); promoted dtype = max(literal, field)
        wdt = np.result_type(lit, fdt).type
        e = dict(lo=(1, 1), hi=(1, 1))
        lap = (lit(4.0).astype(wdt) * gin(0, 0, **e).astype(wdt, copy=False)) - (
            ((gin(1, 0, **e) + gin(-1, 0, **e)) + gin(0, 1, **e)) + gin(0, -1, **e)
        ).astype(wdt, copy=False)
        # lap[a, b] holds the value at domain point (a-1, b-1)
        di, dj, _ = domain

        def glap(oi, oj, lo, hi):
            return lap[1 - lo[0] + oi: 1 + di + hi[0] + oi, 1 - lo[1] + oj: 1 + dj + hi[1] + oj, :]

        # flx over I(-1,0) x J(0,0)
        lo, hi = (1, 0), (0, 0)
        res = glap(1, 0, lo, hi) - glap(0, 0, lo, hi)
        if limiter:
            din = (gin(1, 0, lo, hi) - gin(0, 0, lo, hi)).astype(wdt, copy=False)
            flx = np.where((res * din) > wdt(0), wdt(0), res)
        else:
            flx = res
        # fly over I(0,0) x J(-1,0)
        lo, hi = (0, 1), (0, 0)
        res = glap(0, 1, lo, hi) - glap(0, 0, lo, hi)
        if limiter:
            din = (gin(0, 1, lo, hi) - gin(0, 0, lo, hi)).astype(wdt, copy=False)
            fly = np.where((res * din) > wdt(0), wdt(0), res)
        else:
            fly = res
        # out over the compute domain: flx[a,b] is point (a-1,b), fly[a,b] is point (a,b-1)
        s = ((flx[1:, :, :] - flx[:-1, :, :]) + fly[:, 1:, :]) - fly[:, :-1, :]
        if np.ndim(coeff) == 0:
            cdt = np.asarray(coeff).dtype.type
            c = cdt(coeff)
        else:
            c = _win(coeff, origin_coeff if origin_coeff is not None else origin_in, domain)
            cdt = c.dtype.type
        pdt = np.result_type(cdt, wdt).type
        prod = (c.astype(pdt, copy=False) if np.ndim(c) else pdt(c)) * s.astype(pdt, copy=False)
        rdt = np.result_type(fdt, pdt).type
        r = gin(0, 0).astype(rdt, copy=False) - prod.astype(rdt, copy=False)
        _win(out_field, origin_out, domain)[...] = r.astype(out_field.dtype, copy=False)
    return out_field


def hdiff_validation(u, weight):
    """The reference's own numpy validation of limiter-free horizontal diffusion.

    Restates TestHorizontalDiffusion.validation (tests/cartesian_tests/integration_tests/
    multi_feature_tests/test_suites.py:222-230): ``u`` carries a halo of 2 in I and J, the
    result has the shape of the compute domain.  Used to pin ``hdiff(limiter=False)``.
    """
    c = u[1:-1, 1:-1, :]
    lap = 4.0 * c - (u[2:, 1:-1, :] + u[:-2, 1:-1, :] + u[1:-1, 2:, :] + u[1:-1, :-2, :])
    fi = lap[1:, 1:-1, :] - lap[:-1, 1:-1, :]
    fj = lap[1:-1, 1:, :] - lap[1:-1, :-1, :]
    return u[2:-2, 2:-2, :] - weight * (fi[1:, :, :] - fi[:-1, :, :] + fj[:, 1:, :] - fj[:, :-1, :])


# --------------------------------------------------------------------------------------
# Tridiagonal (Thomas) solver (stencil_definitions.py:219-232)
# --------------------------------------------------------------------------------------
def tridiag(inf, diag, sup, rhs, out, *, origins=None, domain=None):
    """Vertical tridiagonal solve; mutates ``sup`` and ``rhs`` in place as the reference does.

    Sequential K is a Python loop over single-level slices ``k_:k_+1`` (npir_codegen.py:64-71,
    243-248); within a level the statements run in program order over the whole IJ domain
    (SURVEY Appendix A.3).  The denominator ``diag - sup[k-1]*inf`` is evaluated in both
    statements of interval(1, None); the first statement writes ``sup[k]`` only, so both
    evaluations see the same ``sup[k-1]``.
    """
    names = ("inf", "diag", "sup", "rhs", "out")
    arrs = dict(zip(names, (inf, diag, sup, rhs, out)))
    if origins is None:
        origins = {n: (0, 0, 0) for n in names}
    if domain is None:
        domain = inf.shape
    di, dj, dk = domain

    def lvl(name, k):
        oi, oj, ok = origins[name]
        return arrs[name][oi: oi + di, oj: oj + dj, ok + k: ok + k + 1]

    with np.errstate(divide="ignore", over="ignore", under="ignore", invalid="ignore"):
        # FORWARD, interval(0, 1)
        lvl("sup", 0)[...] = lvl("sup", 0) / lvl("diag", 0)
        lvl("rhs", 0)[...] = lvl("rhs", 0) / lvl("diag", 0)
        # FORWARD, interval(1, None)
        for k in range(1, dk):
            lvl("sup", k)[...] = lvl("sup", k) / (lvl("diag", k) - (lvl("sup", k - 1) * lvl("inf", k)))
            lvl("rhs", k)[...] = (lvl("rhs", k) - (lvl("inf", k) * lvl("rhs", k - 1))) / (
                lvl("diag", k) - (lvl("sup", k - 1) * lvl("inf", k))
            )
        # BACKWARD, interval(-1, None)
        lvl("out", dk - 1)[...] = lvl("rhs", dk - 1)
        # BACKWARD, interval(0, -1)
        for k in range(dk - 2, -1, -1):
            lvl("out", k)[...] = lvl("rhs", k) - (lvl("sup", k) * lvl("out", k + 1))
    return out


# --------------------------------------------------------------------------------------
# vertical_advection_dycore (stencil_definitions.py:235-313) -- SURVEY.md section 8f rank 1
# --------------------------------------------------------------------------------------
def vadv(utens_stage, u_stage, wcon, u_pos, utens, dtr_stage, *, origins=None, domain=None, bet_m=0.5, bet_p=0.5):
    """``vertical_advection_dycore``: implicit vertical advection, a Thomas solve whose coefficients are assembled on the
    fly.  Writes ``utens_stage`` in place; every other field is read only.

    Written against the DEFINITION (stencil_definitions.py:235-313), statement by statement, in the numpy backend's
    schedule: sequential K is a Python loop over single-level slices, within a level the statements run in program order
    over the whole IJ domain (npir_codegen.py:64-71, 243-248); three FORWARD interval blocks (0, 1), (1, -1), (-1, None),
    then two BACKWARD blocks (-1, None), (0, -1).  Expressions are evaluated exactly as Python parses them: ``-cs * x`` is
    ``(-cs) * x``, ``a + b + c + d`` associates to the left, ``-0.25 * (...)`` is ``(-(0.25)) * (...)``
    (frontend/gtscript_frontend.py:1477-1504).  All fields are float64 (``Field3D``), the externals BET_M = BET_P = 0.5
    and the literals are float64 (definitions.py:40-42), ``dtr_stage`` is a float64 scalar.  Temporaries (gcv, cs, ccol,
    bcol, correction_term, dcol, divided, gav, as_, acol, datacol) are domain-sized arrays as in npir_codegen.py:88-104.
    ``wcon`` is read at [1, 0, 0], [1, 0, 1] and [0, 0, 1]: it needs one more column and one more level than the domain.
    This function does NOT go through this repository's frontend or IR."""
    names = ("utens_stage", "u_stage", "wcon", "u_pos", "utens")
    arrs = dict(zip(names, (utens_stage, u_stage, wcon, u_pos, utens)))
    if origins is None:
        origins = {n: (0, 0, 0) for n in names}
    if domain is None:
        domain = (wcon.shape[0] - 1, wcon.shape[1], wcon.shape[2] - 1)
    di, dj, dk = domain
    f8 = np.float64
    dtr, BET_M, BET_P = f8(dtr_stage), f8(bet_m), f8(bet_p)

    def lvl(name, k, di_=0, dk_=0):
        oi, oj, ok = origins[name]
        return arrs[name][oi + di_: oi + di_ + di, oj: oj + dj, ok + k + dk_]

    tmp = {n: np.zeros((di, dj, dk), dtype=f8) for n in
           ("gcv", "cs", "ccol", "bcol", "correction_term", "dcol", "divided", "gav", "as_", "acol", "datacol")}
    t = tmp.__getitem__
    with np.errstate(divide="ignore", over="ignore", under="ignore", invalid="ignore"):
        # FORWARD, interval(0, 1)
        k = 0
        t("gcv")[:, :, k] = f8(0.25) * (lvl("wcon", k, 1, 1) + lvl("wcon", k, 0, 1))
        t("cs")[:, :, k] = t("gcv")[:, :, k] * BET_M
        t("ccol")[:, :, k] = t("gcv")[:, :, k] * BET_P
        t("bcol")[:, :, k] = dtr - t("ccol")[:, :, k]
        t("correction_term")[:, :, k] = (-t("cs")[:, :, k]) * (lvl("u_stage", k, 0, 1) - lvl("u_stage", k))
        t("dcol")[:, :, k] = (((dtr * lvl("u_pos", k)) + lvl("utens", k)) + lvl("utens_stage", k)) + t("correction_term")[:, :, k]
        t("divided")[:, :, k] = f8(1.0) / t("bcol")[:, :, k]
        t("ccol")[:, :, k] = t("ccol")[:, :, k] * t("divided")[:, :, k]
        t("dcol")[:, :, k] = t("dcol")[:, :, k] * t("divided")[:, :, k]
        # FORWARD, interval(1, -1)
        for k in range(1, dk - 1):
            t("gav")[:, :, k] = (-f8(0.25)) * (lvl("wcon", k, 1, 0) + lvl("wcon", k))
            t("gcv")[:, :, k] = f8(0.25) * (lvl("wcon", k, 1, 1) + lvl("wcon", k, 0, 1))
            t("as_")[:, :, k] = t("gav")[:, :, k] * BET_M
            t("cs")[:, :, k] = t("gcv")[:, :, k] * BET_M
            t("acol")[:, :, k] = t("gav")[:, :, k] * BET_P
            t("ccol")[:, :, k] = t("gcv")[:, :, k] * BET_P
            t("bcol")[:, :, k] = (dtr - t("acol")[:, :, k]) - t("ccol")[:, :, k]
            t("correction_term")[:, :, k] = ((-t("as_")[:, :, k]) * (lvl("u_stage", k, 0, -1) - lvl("u_stage", k))) - (
                t("cs")[:, :, k] * (lvl("u_stage", k, 0, 1) - lvl("u_stage", k)))
            t("dcol")[:, :, k] = (((dtr * lvl("u_pos", k)) + lvl("utens", k)) + lvl("utens_stage", k)) + t("correction_term")[:, :, k]
            t("divided")[:, :, k] = f8(1.0) / (t("bcol")[:, :, k] - (t("ccol")[:, :, k - 1] * t("acol")[:, :, k]))
            t("ccol")[:, :, k] = t("ccol")[:, :, k] * t("divided")[:, :, k]
            t("dcol")[:, :, k] = (t("dcol")[:, :, k] - (t("dcol")[:, :, k - 1] * t("acol")[:, :, k])) * t("divided")[:, :, k]
        # FORWARD, interval(-1, None)
        k = dk - 1
        t("gav")[:, :, k] = (-f8(0.25)) * (lvl("wcon", k, 1, 0) + lvl("wcon", k))
        t("as_")[:, :, k] = t("gav")[:, :, k] * BET_M
        t("acol")[:, :, k] = t("gav")[:, :, k] * BET_P
        t("bcol")[:, :, k] = dtr - t("acol")[:, :, k]
        t("correction_term")[:, :, k] = (-t("as_")[:, :, k]) * (lvl("u_stage", k, 0, -1) - lvl("u_stage", k))
        t("dcol")[:, :, k] = (((dtr * lvl("u_pos", k)) + lvl("utens", k)) + lvl("utens_stage", k)) + t("correction_term")[:, :, k]
        t("divided")[:, :, k] = f8(1.0) / (t("bcol")[:, :, k] - (t("ccol")[:, :, k - 1] * t("acol")[:, :, k]))
        t("dcol")[:, :, k] = (t("dcol")[:, :, k] - (t("dcol")[:, :, k - 1] * t("acol")[:, :, k])) * t("divided")[:, :, k]
        # BACKWARD, interval(-1, None)
        k = dk - 1
        t("datacol")[:, :, k] = t("dcol")[:, :, k]
        lvl("utens_stage", k)[...] = dtr * (t("datacol")[:, :, k] - lvl("u_pos", k))
        # BACKWARD, interval(0, -1)
        for k in range(dk - 2, -1, -1):
            t("datacol")[:, :, k] = t("dcol")[:, :, k] - (t("ccol")[:, :, k] * t("datacol")[:, :, k + 1])
            lvl("utens_stage", k)[...] = dtr * (t("datacol")[:, :, k] - lvl("u_pos", k))
    return utens_stage


# --------------------------------------------------------------------------------------
# Point-by-point restatements (the "debug" backend's loop order, gtc/debug/debug_codegen.py:
# 93-134).  Pure Python loops: small cases only.  Independent of the slicing code above.
# --------------------------------------------------------------------------------------
def laplacian_loops(inp, out, *, origin_inp, origin_out, domain):
    oi, oj, ok = origin_inp
    pi, pj, pk = origin_out
    f = inp.dtype.type
    for i in range(domain[0]):
        for j in range(domain[1]):
            for k in range(domain[2]):
                a, b, c = oi + i, oj + j, ok + k
                v = f(-4.0) * inp[a, b, c]
                v = v + inp[a - 1, b, c]
                v = v + inp[a + 1, b, c]
                v = v + inp[a, b - 1, c]
                v = v + inp[a, b + 1, c]
                out[pi + i, pj + j, pk + k] = v
    return out


def hdiff_loops(in_field, out_field, coeff, *, origin_in, origin_out, origin_coeff, domain,
                limiter=True):
    """float64-internal point-wise hdiff (default literal precision)."""
    oi, oj, ok = origin_in
    w = np.float64

    def lap(a, b, c):
        s = in_field[a + 1, b, c] + in_field[a - 1, b, c]
        s = s + in_field[a, b + 1, c]
        s = s + in_field[a, b - 1, c]
        return w(4.0) * w(in_field[a, b, c]) - w(s)

    def flux(a, b, c, da, db):
        res = lap(a + da, b + db, c) - lap(a, b, c)
        if limiter:
            d = w(in_field[a + da, b + db, c] - in_field[a, b, c])
            return w(0) if res * d > 0 else res
        return res

    for i in range(domain[0]):
        for j in range(domain[1]):
            for k in range(domain[2]):
                a, b, c = oi + i, oj + j, ok + k
                s = flux(a, b, c, 1, 0) - flux(a - 1, b, c, 1, 0)
                s = s + flux(a, b, c, 0, 1)
                s = s - flux(a, b - 1, c, 0, 1)
                if np.ndim(coeff) == 0:
                    cf = coeff
                else:
                    cf = coeff[origin_coeff[0] + i, origin_coeff[1] + j, origin_coeff[2] + k]
                r = w(in_field[a, b, c]) - w(cf) * s
                out_field[origin_out[0] + i, origin_out[1] + j, origin_out[2] + k] = r
    return out_field


def tridiag_loops(inf, diag, sup, rhs, out):
    """Column-by-column Thomas recurrence (what a thread-per-column kernel executes)."""
    ni, nj, nk = inf.shape
    for i in range(ni):
        for j in range(nj):
            sup[i, j, 0] = sup[i, j, 0] / diag[i, j, 0]
            rhs[i, j, 0] = rhs[i, j, 0] / diag[i, j, 0]
            for k in range(1, nk):
                den = diag[i, j, k] - sup[i, j, k - 1] * inf[i, j, k]
                sup[i, j, k] = sup[i, j, k] / den
                rhs[i, j, k] = (rhs[i, j, k] - inf[i, j, k] * rhs[i, j, k - 1]) / den
            out[i, j, nk - 1] = rhs[i, j, nk - 1]
            for k in range(nk - 2, -1, -1):
                out[i, j, k] = rhs[i, j, k] - sup[i, j, k] * out[i, j, k + 1]
    return out
