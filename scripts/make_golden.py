"""Generate the golden fixtures under tests/golden/ (run in the build container only).

Two pieces of the reference ARE loadable stand-alone under Python 3.10:

* /root/reference/src/gt4py/storage/cartesian/layout.py  (numpy + typing only)
  -> tests/golden/layout_tables.json: ``layout_maker_factory(base)(dims)`` for every base layout
     and dimension tuple the reference's own test tables cover
     (/root/reference/tests/storage_tests/unit_tests/test_layout.py:16-131), plus ``_check_layout``
     verdicts for a set of stride tuples.  These are outputs of the reference code itself.

* /root/reference/src/gt4py/cartesian/utils/field.py  (the origin-shifting ``Field`` shim that every
  numpy-backend generated module uses)
  -> tests/golden/stencils_small.npz: the three hot-path stencils executed as the statement-level
     numpy code of SURVEY.md Appendix A (hand-derived from the codegen rules; the package itself
     cannot be imported here) ON TOP OF the reference's real ``Field`` class.  Inputs are seeded;
     inputs and outputs are stored.  This pins origin/extent/slice arithmetic to reference code and
     the arithmetic to numpy; it is "reference-assisted", not a capture of generated code.

Nothing here runs on the GPU box and nothing under /root/reference is copied into the repo.
"""

from __future__ import annotations

import importlib.util
import itertools
import json
import pathlib
import sys

import numpy as np

REF = pathlib.Path("/root/reference/src/gt4py")
OUT = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden"


def _load(path: pathlib.Path, name: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_layout_tables() -> None:
    layout = _load(REF / "storage" / "cartesian" / "layout.py", "_ref_layout")
    carts = [c for n in range(4) for c in itertools.combinations("IJK", n)]
    datas = [(), ("0",), ("1",), ("0", "1")]
    tables = {}
    for base in [(0, 1, 2), (2, 0, 1), (2, 1, 0), (1, 2, 0)]:
        maker = layout.layout_maker_factory(base)
        rows = []
        for cart in carts:
            for data in datas:
                dims = tuple(cart) + tuple(data)
                rows.append([list(dims), [int(v) for v in maker(dims)]])
        tables[",".join(map(str, base))] = rows
    checks = []
    for lm, strides in [((2, 1, 0), (8, 80, 800)), ((2, 1, 0), (800, 80, 8)), ((0, 1, 2), (800, 80, 8)),
                        ((0, 1, 2), (8, 80, 800)), ((2, 1, 0), (8, 8, 800)), ((2, 0, 1), (8, 800, 80)),
                        ((1, 0), (8, 80)), ((0, 1), (8, 80)), ((2, 1, 0), (8, 80)), ((0,), (8,))]:
        checks.append([list(lm), list(strides), bool(layout._check_layout(lm, strides))])
    (OUT / "layout_tables.json").write_text(json.dumps({"layout_maker": tables, "check_layout": checks}, indent=1))


def make_stencil_vectors() -> None:
    Field = _load(REF / "cartesian" / "utils" / "field.py", "_ref_field").Field
    T3 = (True, True, True)
    rng = np.random.default_rng(20261001)
    out = {}

    # ---- A.1 Laplacian (examples/lap_cartesian_vs_next.ipynb cell 7), origins differ per field ----
    dom = (7, 5, 3)
    inp_a = rng.uniform(-1, 1, (10, 9, 4))
    out_a = rng.uniform(-1, 1, (9, 8, 5))
    o_inp, o_out = (2, 1, 1), (1, 3, 2)
    res = out_a.copy()
    inp, outf = Field(inp_a, o_inp, T3), Field(res, o_out, T3)
    i, I, j, J, k, K = 0, dom[0], 0, dom[1], 0, dom[2]
    outf[i:I, j:J, k:K] = ((((((-(np.float64(4.0))) * inp[i:I, j:J, k:K]) + inp[i - 1:I - 1, j:J, k:K])
                             + inp[i + 1:I + 1, j:J, k:K]) + inp[i:I, j - 1:J - 1, k:K]) + inp[i:I, j + 1:J + 1, k:K])
    out.update(lap_inp=inp_a, lap_out0=out_a, lap_out=res, lap_origin_inp=o_inp, lap_origin_out=o_out, lap_domain=dom)

    # ---- A.2 horizontal diffusion with limiter, f64 and f32 fields (default f64 literals) --------
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        dom = (6, 7, 2)
        in_a = rng.uniform(-10, 10, (11, 12, 3)).astype(dt)
        cf_a = rng.uniform(0, 0.5, (8, 9, 2)).astype(dt)
        out_a = np.zeros((7, 8, 4), dt)
        o_in, o_cf, o_out = (3, 2, 1), (1, 1, 0), (0, 1, 2)
        res_a = out_a.copy()
        fin, fcf, fout = Field(in_a, o_in, T3), Field(cf_a, o_cf, T3), Field(res_a, o_out, T3)
        dI, dJ, dK = dom
        W = np.float64
        lap = Field.empty((dI + 2, dJ + 2, dK), W, (1, 1, 0), T3)
        flx = Field.empty((dI + 1, dJ, dK), W, (1, 0, 0), T3)
        fly = Field.empty((dI, dJ + 1, dK), W, (0, 1, 0), T3)
        res = Field.empty((dI + 1, dJ + 1, dK), W, (1, 1, 0), T3)
        k, K = 0, dK
        i, I, j, J = -1, dI + 1, -1, dJ + 1
        lap[i:I, j:J, k:K] = (np.float64(4.0) * fin[i:I, j:J, k:K].astype(W)) - (
            ((fin[i + 1:I + 1, j:J, k:K] + fin[i - 1:I - 1, j:J, k:K]) + fin[i:I, j + 1:J + 1, k:K])
            + fin[i:I, j - 1:J - 1, k:K]).astype(W)
        i, I, j, J = -1, dI, 0, dJ
        res[i:I, j:J, k:K] = lap[i + 1:I + 1, j:J, k:K] - lap[i:I, j:J, k:K]
        flx[i:I, j:J, k:K] = np.where(
            (res[i:I, j:J, k:K] * (fin[i + 1:I + 1, j:J, k:K] - fin[i:I, j:J, k:K]).astype(W)) > np.float64(np.int64(0)),
            np.float64(np.int64(0)), res[i:I, j:J, k:K])
        i, I, j, J = 0, dI, -1, dJ
        res[i:I, j:J, k:K] = lap[i:I, j + 1:J + 1, k:K] - lap[i:I, j:J, k:K]
        fly[i:I, j:J, k:K] = np.where(
            (res[i:I, j:J, k:K] * (fin[i:I, j + 1:J + 1, k:K] - fin[i:I, j:J, k:K]).astype(W)) > np.float64(np.int64(0)),
            np.float64(np.int64(0)), res[i:I, j:J, k:K])
        i, I, j, J = 0, dI, 0, dJ
        fout[i:I, j:J, k:K] = (fin[i:I, j:J, k:K].astype(W) - (fcf[i:I, j:J, k:K].astype(W) * (
            ((flx[i:I, j:J, k:K] - flx[i - 1:I - 1, j:J, k:K]) + fly[i:I, j:J, k:K]) - fly[i:I, j - 1:J - 1, k:K]))).astype(dt)
        out.update({f"hd_{tag}_in": in_a, f"hd_{tag}_coeff": cf_a, f"hd_{tag}_out0": out_a, f"hd_{tag}_out": res_a,
                    f"hd_{tag}_origins": np.array([o_in, o_cf, o_out]), f"hd_{tag}_domain": dom})

    # ---- A.3 tridiagonal solver -------------------------------------------------------------------
    dom = (4, 3, 9)
    shp = (5, 4, 9)
    a_inf, a_diag = rng.uniform(-1, 1, shp), rng.uniform(4, 5, shp)
    a_sup, a_rhs = rng.uniform(-1, 1, shp), rng.uniform(-10, 10, shp)
    r_sup, r_rhs, r_out = a_sup.copy(), a_rhs.copy(), np.zeros(shp)
    o = (1, 1, 0)
    inf, diag, sup, rhs, outf = (Field(x, o, T3) for x in (a_inf, a_diag, r_sup, r_rhs, r_out))
    i, I, j, J = 0, dom[0], 0, dom[1]
    dK = dom[2]
    for k_ in range(0, 1):
        sup[i:I, j:J, k_:k_ + 1] = sup[i:I, j:J, k_:k_ + 1] / diag[i:I, j:J, k_:k_ + 1]
        rhs[i:I, j:J, k_:k_ + 1] = rhs[i:I, j:J, k_:k_ + 1] / diag[i:I, j:J, k_:k_ + 1]
    for k_ in range(1, dK):
        sup[i:I, j:J, k_:k_ + 1] = sup[i:I, j:J, k_:k_ + 1] / (
            diag[i:I, j:J, k_:k_ + 1] - (sup[i:I, j:J, k_ - 1:k_] * inf[i:I, j:J, k_:k_ + 1]))
        rhs[i:I, j:J, k_:k_ + 1] = (rhs[i:I, j:J, k_:k_ + 1] - (inf[i:I, j:J, k_:k_ + 1] * rhs[i:I, j:J, k_ - 1:k_])) / (
            diag[i:I, j:J, k_:k_ + 1] - (sup[i:I, j:J, k_ - 1:k_] * inf[i:I, j:J, k_:k_ + 1]))
    for k_ in range(dK - 1, dK - 2, -1):
        outf[i:I, j:J, k_:k_ + 1] = rhs[i:I, j:J, k_:k_ + 1]
    for k_ in range(dK - 2, -1, -1):
        outf[i:I, j:J, k_:k_ + 1] = rhs[i:I, j:J, k_:k_ + 1] - (sup[i:I, j:J, k_:k_ + 1] * outf[i:I, j:J, k_ + 1:k_ + 2])
    out.update(tri_inf=a_inf, tri_diag=a_diag, tri_sup0=a_sup, tri_rhs0=a_rhs, tri_sup=r_sup, tri_rhs=r_rhs,
               tri_out=r_out, tri_origin=o, tri_domain=dom)

    np.savez_compressed(OUT / "stencils_small.npz", **out)


def make_vadv_vector() -> None:
    """tests/golden/vadv_small.npz: ``vertical_advection_dycore`` (stencil_definitions.py:235-313) as the statement code
    the numpy backend's rules give (sequential K = single-level slices k_:k_+1, statements in program order, temporaries as
    domain-sized ``Field.empty`` arrays, npir_codegen.py:64-104, 205-212, 243-248) on the reference's real ``Field`` shim,
    with a different origin per field.  A file of its own: stencils_small.npz stays byte for byte what it was."""
    Field = _load(REF / "cartesian" / "utils" / "field.py", "_ref_field").Field
    T3 = (True, True, True)
    rng = np.random.default_rng(20261002)
    dom = (5, 4, 11)
    dI, dJ, dK = dom
    f8 = np.float64
    origins = {"utens_stage": (1, 0, 2), "u_stage": (0, 2, 1), "wcon": (2, 1, 0), "u_pos": (0, 0, 0), "utens": (1, 1, 1)}
    shapes = {"utens_stage": (7, 5, 14), "u_stage": (6, 7, 13), "wcon": (8, 6, 12), "u_pos": (5, 4, 11), "utens": (6, 6, 13)}
    arrays = {n: rng.uniform(-1, 1, shapes[n]) for n in origins}
    result = arrays["utens_stage"].copy()
    utens_stage, u_stage, wcon, u_pos, utens = (
        Field(result if n == "utens_stage" else arrays[n], origins[n], T3) for n in ("utens_stage", "u_stage", "wcon", "u_pos", "utens"))
    dtr_stage, BET_M, BET_P = f8(3.0 / 20.0), f8(0.5), f8(0.5)
    tmp = {n: Field.empty((dI, dJ, dK), f8, (0, 0, 0), T3) for n in
           ("gcv", "cs", "ccol", "bcol", "correction_term", "dcol", "divided", "gav", "as_", "acol", "datacol")}
    gcv, cs, ccol, bcol, correction_term, dcol, divided, gav, as_, acol, datacol = (tmp[n] for n in (
        "gcv", "cs", "ccol", "bcol", "correction_term", "dcol", "divided", "gav", "as_", "acol", "datacol"))
    i, I, j, J = 0, dI, 0, dJ
    for k_ in range(0, 1):  # FORWARD, interval(0, 1)
        s, n1 = slice(k_, k_ + 1), slice(k_ + 1, k_ + 2)
        gcv[i:I, j:J, s] = f8(0.25) * (wcon[i + 1:I + 1, j:J, n1] + wcon[i:I, j:J, n1])
        cs[i:I, j:J, s] = gcv[i:I, j:J, s] * BET_M
        ccol[i:I, j:J, s] = gcv[i:I, j:J, s] * BET_P
        bcol[i:I, j:J, s] = dtr_stage - ccol[i:I, j:J, s]
        correction_term[i:I, j:J, s] = (-cs[i:I, j:J, s]) * (u_stage[i:I, j:J, n1] - u_stage[i:I, j:J, s])
        dcol[i:I, j:J, s] = (((dtr_stage * u_pos[i:I, j:J, s]) + utens[i:I, j:J, s]) + utens_stage[i:I, j:J, s]) + correction_term[i:I, j:J, s]
        divided[i:I, j:J, s] = f8(1.0) / bcol[i:I, j:J, s]
        ccol[i:I, j:J, s] = ccol[i:I, j:J, s] * divided[i:I, j:J, s]
        dcol[i:I, j:J, s] = dcol[i:I, j:J, s] * divided[i:I, j:J, s]
    for k_ in range(1, dK - 1):  # FORWARD, interval(1, -1)
        s, n1, p1 = slice(k_, k_ + 1), slice(k_ + 1, k_ + 2), slice(k_ - 1, k_)
        gav[i:I, j:J, s] = (-f8(0.25)) * (wcon[i + 1:I + 1, j:J, s] + wcon[i:I, j:J, s])
        gcv[i:I, j:J, s] = f8(0.25) * (wcon[i + 1:I + 1, j:J, n1] + wcon[i:I, j:J, n1])
        as_[i:I, j:J, s] = gav[i:I, j:J, s] * BET_M
        cs[i:I, j:J, s] = gcv[i:I, j:J, s] * BET_M
        acol[i:I, j:J, s] = gav[i:I, j:J, s] * BET_P
        ccol[i:I, j:J, s] = gcv[i:I, j:J, s] * BET_P
        bcol[i:I, j:J, s] = (dtr_stage - acol[i:I, j:J, s]) - ccol[i:I, j:J, s]
        correction_term[i:I, j:J, s] = ((-as_[i:I, j:J, s]) * (u_stage[i:I, j:J, p1] - u_stage[i:I, j:J, s])) - (
            cs[i:I, j:J, s] * (u_stage[i:I, j:J, n1] - u_stage[i:I, j:J, s]))
        dcol[i:I, j:J, s] = (((dtr_stage * u_pos[i:I, j:J, s]) + utens[i:I, j:J, s]) + utens_stage[i:I, j:J, s]) + correction_term[i:I, j:J, s]
        divided[i:I, j:J, s] = f8(1.0) / (bcol[i:I, j:J, s] - (ccol[i:I, j:J, p1] * acol[i:I, j:J, s]))
        ccol[i:I, j:J, s] = ccol[i:I, j:J, s] * divided[i:I, j:J, s]
        dcol[i:I, j:J, s] = (dcol[i:I, j:J, s] - (dcol[i:I, j:J, p1] * acol[i:I, j:J, s])) * divided[i:I, j:J, s]
    for k_ in range(dK - 1, dK):  # FORWARD, interval(-1, None)
        s, p1 = slice(k_, k_ + 1), slice(k_ - 1, k_)
        gav[i:I, j:J, s] = (-f8(0.25)) * (wcon[i + 1:I + 1, j:J, s] + wcon[i:I, j:J, s])
        as_[i:I, j:J, s] = gav[i:I, j:J, s] * BET_M
        acol[i:I, j:J, s] = gav[i:I, j:J, s] * BET_P
        bcol[i:I, j:J, s] = dtr_stage - acol[i:I, j:J, s]
        correction_term[i:I, j:J, s] = (-as_[i:I, j:J, s]) * (u_stage[i:I, j:J, p1] - u_stage[i:I, j:J, s])
        dcol[i:I, j:J, s] = (((dtr_stage * u_pos[i:I, j:J, s]) + utens[i:I, j:J, s]) + utens_stage[i:I, j:J, s]) + correction_term[i:I, j:J, s]
        divided[i:I, j:J, s] = f8(1.0) / (bcol[i:I, j:J, s] - (ccol[i:I, j:J, p1] * acol[i:I, j:J, s]))
        dcol[i:I, j:J, s] = (dcol[i:I, j:J, s] - (dcol[i:I, j:J, p1] * acol[i:I, j:J, s])) * divided[i:I, j:J, s]
    for k_ in range(dK - 1, dK - 2, -1):  # BACKWARD, interval(-1, None)
        s = slice(k_, k_ + 1)
        datacol[i:I, j:J, s] = dcol[i:I, j:J, s]
        utens_stage[i:I, j:J, s] = dtr_stage * (datacol[i:I, j:J, s] - u_pos[i:I, j:J, s])
    for k_ in range(dK - 2, -1, -1):  # BACKWARD, interval(0, -1)
        s, n1 = slice(k_, k_ + 1), slice(k_ + 1, k_ + 2)
        datacol[i:I, j:J, s] = dcol[i:I, j:J, s] - (ccol[i:I, j:J, s] * datacol[i:I, j:J, n1])
        utens_stage[i:I, j:J, s] = dtr_stage * (datacol[i:I, j:J, s] - u_pos[i:I, j:J, s])
    out = {f"vadv_{n}": arrays[n] for n in arrays}
    out.update(vadv_utens_stage_out=result, vadv_domain=dom, vadv_dtr_stage=float(dtr_stage),
               vadv_origins=np.array([origins[n] for n in ("utens_stage", "u_stage", "wcon", "u_pos", "utens")]))
    np.savez_compressed(OUT / "vadv_small.npz", **out)


if __name__ == "__main__":
    if not REF.exists():
        sys.exit("the reference tree is not available here; fixtures are committed under tests/golden/")
    OUT.mkdir(parents=True, exist_ok=True)
    make_layout_tables()
    make_stencil_vectors()
    make_vadv_vector()
    print("wrote", sorted(p.name for p in OUT.iterdir()))
