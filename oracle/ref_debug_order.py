"""ORACLE (test infrastructure, NOT product code): a third, structurally different restatement of the three
hot stencils -- scalar arithmetic point by point in the loop order of the reference's ``debug`` backend.

/root/reference/src/gt4py/cartesian/gtc/debug/debug_codegen.py:93-124 emits, per vertical loop, per interval
section and per horizontal execution (= one statement here, no merging passes run for that backend)

    for i in range(i_0 + ext_i_lo, i_size + ext_i_hi):
        for j in range(j_0 + ext_j_lo, j_size + ext_j_hi):
            for k in range(interval, +-1):
                <statement at the single point (i, j, k)>

i.e. IJ outermost, K innermost, one statement at a time over its own block extent -- neither the numpy
backend's whole-array slices (``ref_numpy.laplacian / hdiff / tridiag``) nor a column-at-a-time recurrence
(``ref_numpy.*_loops``).  Every operation is one numpy-scalar operation of the dtype the reference's upcasting
rules give it (gtir_upcaster.py:43-143), so each rounding happens exactly where the other restatements have it;
tests/test_oracle_restatements.py asserts that all three agree bit for bit and with the golden vectors.

Pure Python loops: small cases only.
"""

from __future__ import annotations

import numpy as np


def laplacian_debug_order(inp, out, *, origin_inp=(1, 1, 0), origin_out=(1, 1, 0), domain=None,
                          literal_float_precision=64):
    """examples/lap_cartesian_vs_next.ipynb cell 7: out = -4.0*inp + inp[-1,0,0] + inp[1,0,0] + inp[0,-1,0] + inp[0,1,0].
    ``literal_float_precision`` is the build option of that name (gtscript.py:219 ff.): 64 (the default) makes the
    literal float64, which widens every float32 operand it meets; 32 keeps float32 fields in float32 throughout
    (what ``ref_numpy.laplacian`` restates for float32 arrays)."""
    oi, oj, ok = origin_inp
    pi, pj, pk = origin_out
    dI, dJ, dK = domain
    m4 = -(np.float64(4.0) if literal_float_precision == 64 else np.float32(4.0))
    with np.errstate(all="ignore"):
        for i in range(dI):
            for j in range(dJ):
                for k in range(dK):
                    c = inp[oi + i, oj + j, ok + k]
                    v = m4 * c
                    v = v + inp[oi + i - 1, oj + j, ok + k]
                    v = v + inp[oi + i + 1, oj + j, ok + k]
                    v = v + inp[oi + i, oj + j - 1, ok + k]
                    v = v + inp[oi + i, oj + j + 1, ok + k]
                    out[pi + i, pj + j, pk + k] = v


def hdiff_debug_order(in_field, out_field, coeff, *, origin_in=(2, 2, 0), origin_out=(2, 2, 0), origin_coeff=None,
                      domain=None, limiter=True):
    """stencil_definitions.py:316-328 (limiter) / :206-216 (plain), default float64 literals: lap, flx, fly and the
    re-assigned ``res`` are float64 temporaries; float32 operands are widened where they meet a float64 one."""
    oi, oj, ok = origin_in
    pi, pj, pk = origin_out
    ci, cj, ck = origin_coeff if origin_coeff is not None else origin_in
    dI, dJ, dK = domain
    T, W = in_field.dtype.type, np.float64
    four, zero = W(4.0), W(np.int64(0))

    def a(i, j, k):
        return in_field[oi + i, oj + j, ok + k]

    # temporaries over the union of the extents they are written / read on (I: -1..dI, J: -1..dJ), origin (1, 1)
    lap = np.full((dI + 2, dJ + 2, dK), np.nan, W)
    res = np.full((dI + 2, dJ + 2, dK), np.nan, W)
    flx = np.full((dI + 2, dJ + 2, dK), np.nan, W)
    fly = np.full((dI + 2, dJ + 2, dK), np.nan, W)
    with np.errstate(all="ignore"):
        for i in range(-1, dI + 1):  # lap: extent I(-1, 1) x J(-1, 1)
            for j in range(-1, dJ + 1):
                for k in range(dK):
                    s = T(T(T(a(i + 1, j, k) + a(i - 1, j, k)) + a(i, j + 1, k)) + a(i, j - 1, k))
                    lap[i + 1, j + 1, k] = (four * W(a(i, j, k))) - W(s)
        if limiter:
            for i in range(-1, dI):  # res = lap[1,0,0] - lap: the extent of its reader flx, I(-1, 0) x J(0, 0)
                for j in range(0, dJ):
                    for k in range(dK):
                        res[i + 1, j + 1, k] = lap[i + 2, j + 1, k] - lap[i + 1, j + 1, k]
            for i in range(-1, dI):  # flx = 0 if res * (in[1,0,0] - in) > 0 else res
                for j in range(0, dJ):
                    for k in range(dK):
                        r = res[i + 1, j + 1, k]
                        flx[i + 1, j + 1, k] = zero if (r * W(T(a(i + 1, j, k) - a(i, j, k)))) > zero else r
            for i in range(0, dI):  # res = lap[0,1,0] - lap: I(0, 0) x J(-1, 0)
                for j in range(-1, dJ):
                    for k in range(dK):
                        res[i + 1, j + 1, k] = lap[i + 1, j + 2, k] - lap[i + 1, j + 1, k]
            for i in range(0, dI):  # fly
                for j in range(-1, dJ):
                    for k in range(dK):
                        r = res[i + 1, j + 1, k]
                        fly[i + 1, j + 1, k] = zero if (r * W(T(a(i, j + 1, k) - a(i, j, k)))) > zero else r
        else:
            for i in range(-1, dI):  # flx = lap[1,0,0] - lap
                for j in range(0, dJ):
                    for k in range(dK):
                        flx[i + 1, j + 1, k] = lap[i + 2, j + 1, k] - lap[i + 1, j + 1, k]
            for i in range(0, dI):  # fly = lap[0,1,0] - lap
                for j in range(-1, dJ):
                    for k in range(dK):
                        fly[i + 1, j + 1, k] = lap[i + 1, j + 2, k] - lap[i + 1, j + 1, k]
        for i in range(dI):  # out = in - coeff * (flx - flx[-1,0,0] + fly - fly[0,-1,0])
            for j in range(dJ):
                for k in range(dK):
                    s = ((flx[i + 1, j + 1, k] - flx[i, j + 1, k]) + fly[i + 1, j + 1, k]) - fly[i + 1, j, k]
                    c = W(coeff[ci + i, cj + j, ck + k]) if isinstance(coeff, np.ndarray) else W(coeff)
                    out_field[pi + i, pj + j, pk + k] = T(W(a(i, j, k)) - (c * s))


def tridiag_debug_order(inf, diag, sup, rhs, out, *, origins=None, domain=None):
    """stencil_definitions.py:219-232; mutates sup and rhs like the reference.  Statement by statement: the whole
    (i, j, k) sweep of `sup = ...` finishes before the sweep of `rhs = ...` starts, so the denominator of the second
    statement reads the UPDATED sup[k-1] -- the same value the level-by-level evaluation of the numpy backend sees."""
    names = ("inf", "diag", "sup", "rhs", "out")
    origins = origins or {n: (0, 0, 0) for n in names}
    dI, dJ, dK = domain if domain is not None else inf.shape

    def at(arr, name, i, j, k):
        o = origins[name]
        return (o[0] + i, o[1] + j, o[2] + k)

    with np.errstate(all="ignore"):
        # FORWARD interval(0, 1)
        for i in range(dI):
            for j in range(dJ):
                sup[at(sup, "sup", i, j, 0)] = sup[at(sup, "sup", i, j, 0)] / diag[at(diag, "diag", i, j, 0)]
        for i in range(dI):
            for j in range(dJ):
                rhs[at(rhs, "rhs", i, j, 0)] = rhs[at(rhs, "rhs", i, j, 0)] / diag[at(diag, "diag", i, j, 0)]
        # FORWARD interval(1, None)
        for i in range(dI):
            for j in range(dJ):
                for k in range(1, dK):
                    den = diag[at(diag, "diag", i, j, k)] - (sup[at(sup, "sup", i, j, k - 1)] * inf[at(inf, "inf", i, j, k)])
                    sup[at(sup, "sup", i, j, k)] = sup[at(sup, "sup", i, j, k)] / den
        for i in range(dI):
            for j in range(dJ):
                for k in range(1, dK):
                    num = rhs[at(rhs, "rhs", i, j, k)] - (inf[at(inf, "inf", i, j, k)] * rhs[at(rhs, "rhs", i, j, k - 1)])
                    den = diag[at(diag, "diag", i, j, k)] - (sup[at(sup, "sup", i, j, k - 1)] * inf[at(inf, "inf", i, j, k)])
                    rhs[at(rhs, "rhs", i, j, k)] = num / den
        # BACKWARD interval(-1, None), then interval(0, -1)
        for i in range(dI):
            for j in range(dJ):
                out[at(out, "out", i, j, dK - 1)] = rhs[at(rhs, "rhs", i, j, dK - 1)]
        for i in range(dI):
            for j in range(dJ):
                for k in range(dK - 2, -1, -1):
                    out[at(out, "out", i, j, k)] = rhs[at(rhs, "rhs", i, j, k)] - (
                        sup[at(sup, "sup", i, j, k)] * out[at(out, "out", i, j, k + 1)])


def vadv_debug_order(utens_stage, u_stage, wcon, u_pos, utens, dtr_stage, *, origins=None, domain=None, bet_m=0.5, bet_p=0.5):
    """``vertical_advection_dycore`` (stencil_definitions.py:235-313) in the debug backend's order.  That backend merges
    the statements of an interval section into ONE horizontal execution when none reads another's result at a horizontal
    offset (debug_backend.py:42-45: HorizontalExecutionMerging; here only the never-written ``wcon`` is read at an I
    offset), so the generated loops are, per vertical loop and section,

        for i: for j: for k in section (forward or backward): every statement of the section at the point (i, j, k)

    -- a whole IJ sweep per interval section, columns innermost: neither the numpy backend's level-by-level slices
    (``ref_numpy.vadv``) nor one column through all sections at a time.  Temporaries are domain-sized arrays between the
    sections, numpy float64 scalars inside."""
    names = ("utens_stage", "u_stage", "wcon", "u_pos", "utens")
    arrs = dict(zip(names, (utens_stage, u_stage, wcon, u_pos, utens)))
    origins = origins or {n: (0, 0, 0) for n in names}
    dI, dJ, dK = domain if domain is not None else (wcon.shape[0] - 1, wcon.shape[1], wcon.shape[2] - 1)
    f8 = np.float64
    dtr, BET_M, BET_P = f8(dtr_stage), f8(bet_m), f8(bet_p)

    def at(name, i, j, k):
        o = origins[name]
        return arrs[name][o[0] + i, o[1] + j, o[2] + k]

    ccol = np.zeros((dI, dJ, dK))
    dcol = np.zeros((dI, dJ, dK))
    datacol = np.zeros((dI, dJ, dK))
    with np.errstate(all="ignore"):
        for i in range(dI):  # FORWARD, interval(0, 1)
            for j in range(dJ):
                k = 0
                gcv = f8(0.25) * (at("wcon", i + 1, j, k + 1) + at("wcon", i, j, k + 1))
                cs = gcv * BET_M
                c = gcv * BET_P
                bcol = dtr - c
                correction_term = (-cs) * (at("u_stage", i, j, k + 1) - at("u_stage", i, j, k))
                d = (((dtr * at("u_pos", i, j, k)) + at("utens", i, j, k)) + at("utens_stage", i, j, k)) + correction_term
                divided = f8(1.0) / bcol
                ccol[i, j, k] = c * divided
                dcol[i, j, k] = d * divided
        for i in range(dI):  # FORWARD, interval(1, -1)
            for j in range(dJ):
                for k in range(1, dK - 1):
                    gav = (-f8(0.25)) * (at("wcon", i + 1, j, k) + at("wcon", i, j, k))
                    gcv = f8(0.25) * (at("wcon", i + 1, j, k + 1) + at("wcon", i, j, k + 1))
                    as_ = gav * BET_M
                    cs = gcv * BET_M
                    acol = gav * BET_P
                    c = gcv * BET_P
                    bcol = (dtr - acol) - c
                    correction_term = ((-as_) * (at("u_stage", i, j, k - 1) - at("u_stage", i, j, k))) - (
                        cs * (at("u_stage", i, j, k + 1) - at("u_stage", i, j, k)))
                    d = (((dtr * at("u_pos", i, j, k)) + at("utens", i, j, k)) + at("utens_stage", i, j, k)) + correction_term
                    divided = f8(1.0) / (bcol - (ccol[i, j, k - 1] * acol))
                    ccol[i, j, k] = c * divided
                    dcol[i, j, k] = (d - (dcol[i, j, k - 1] * acol)) * divided
        for i in range(dI):  # FORWARD, interval(-1, None)
            for j in range(dJ):
                k = dK - 1
                gav = (-f8(0.25)) * (at("wcon", i + 1, j, k) + at("wcon", i, j, k))
                as_ = gav * BET_M
                acol = gav * BET_P
                bcol = dtr - acol
                correction_term = (-as_) * (at("u_stage", i, j, k - 1) - at("u_stage", i, j, k))
                d = (((dtr * at("u_pos", i, j, k)) + at("utens", i, j, k)) + at("utens_stage", i, j, k)) + correction_term
                divided = f8(1.0) / (bcol - (ccol[i, j, k - 1] * acol))
                dcol[i, j, k] = (d - (dcol[i, j, k - 1] * acol)) * divided
        o = origins["utens_stage"]
        for i in range(dI):  # BACKWARD, interval(-1, None)
            for j in range(dJ):
                k = dK - 1
                datacol[i, j, k] = dcol[i, j, k]
                utens_stage[o[0] + i, o[1] + j, o[2] + k] = dtr * (datacol[i, j, k] - at("u_pos", i, j, k))
        for i in range(dI):  # BACKWARD, interval(0, -1)
            for j in range(dJ):
                for k in range(dK - 2, -1, -1):
                    datacol[i, j, k] = dcol[i, j, k] - (ccol[i, j, k] * datacol[i, j, k + 1])
                    utens_stage[o[0] + i, o[1] + j, o[2] + k] = dtr * (datacol[i, j, k] - at("u_pos", i, j, k))
    return utens_stage

