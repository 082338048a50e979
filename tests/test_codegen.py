"""CPU: the generic executor's planner and code generator (gt4py_amd/cartesian/backend/hip_codegen.py).

No kernel runs here.  Checked without a GPU: (1) the stage plan of each zoo stencil; (2) the rewritten
(inlined) IR evaluates to exactly the same values as the original under the numpy oracle; (3) every
generated program compiles for gfx950 through the library's hiprtc entry point, in both layout
variants.  Values on the device are checked in tests/test_gpu_generic.py.
"""

import numpy as np
import pytest

import oracle.numpy_backend as oracle_backend  # registers backend "numpy"
import stencil_zoo as zoo
from gt4py_amd import _lib
from gt4py_amd.cartesian import analysis, gtscript, ir
from gt4py_amd.cartesian.backend import hip_codegen


def _build(name, backend):
    defn, externals, _, opts = zoo.ZOO[name]
    return gtscript.stencil(backend=backend, definition=defn, externals=externals,
                            **(opts if backend != "numpy" else {}))


@pytest.fixture(scope="module")
def programs():
    return {name: type(_build(name, "hip:mi300"))._gt_program_ for name in zoo.ZOO}


def test_every_zoo_stencil_goes_to_the_generic_executor(programs):
    assert all(isinstance(p, hip_codegen.GeneratedProgram) for p in programs.values())
    # and without the option the kernel library still wins for its own shapes
    for name in ("horizontal_diffusion", "laplacian", "tridiagonal_solver"):
        defn, externals, _, _ = zoo.ZOO[name]
        obj = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals)
        assert hasattr(type(obj), "_gt_binding_") and not hasattr(type(obj), "_gt_program_")


EXPECTED_PLANS = {
    # name: ([(mapping, extent)], scratch, register_only, forwarded names)
    "copy_stencil": ([("ijk", ((0, 0), (0, 0)))], [], [], []),
    "native_functions": ([("ijk", ((0, 0), (0, 0)))], [], [], []),
    "k_intervals": ([("ijk", ((0, 0), (0, 0)))], [], [], []),
    "horizontal_diffusion": ([("ijk", ((0, 0), (0, 0)))], [], [], []),
    "tridiagonal_solver": ([("column", ((0, 0), (0, 0)))], [], [], ["out", "rhs", "sup"]),
    "vertical_advection_dycore": ([("column", ((0, 0), (0, 0)))], ["ccol", "dcol"], ["datacol"],
                                  ["ccol", "datacol", "dcol"]),
    "column_sum_then_gradient": ([("column", ((-1, 1), (0, 1))), ("ijk", ((0, 0), (0, 0)))], ["acc"], [], ["acc"]),
    "backward_scan": ([("column", ((0, 0), (0, 0)))], [], [], ["out"]),
    "parallel_k_dependency": ([("column", ((0, 0), (0, 0)))], ["tmp"], [], []),
    "two_stage_written_input": ([("ijk", ((0, 1), (-1, 0))), ("ijk", ((0, 0), (0, 0)))], ["t"], [], []),
}


@pytest.mark.parametrize("name", sorted(EXPECTED_PLANS))
def test_stage_plan(programs, name):
    plan = programs[name].plan
    stages, scratch, register_only, forwarded = EXPECTED_PLANS[name]
    assert [(s.mapping, s.extent) for s in plan.stages] == stages
    assert sorted(plan.scratch) == scratch
    assert sorted(plan.register_only) == register_only
    assert sorted({n for _, n in plan.forwarded}) == forwarded
    assert not programs[name].inexact_calls


def test_hdiff_is_recomputed_not_staged(programs):
    """lap/flx/fly disappear as fields: one kernel, no scratch, and the input is read up to 2 cells away."""
    plan = programs["horizontal_diffusion"].plan
    offsets = {e.offset[:2] for _, _, s in plan.stencil.statements() for e in ir.walk(s.value)
               if isinstance(e, ir.FieldAccess) and e.name == "in_field"}
    assert {(2, 0), (0, 2), (-2, 0), (0, -2), (1, 1), (-1, 1), (1, -1), (-1, -1), (0, 0)} <= offsets
    assert max(abs(i) + abs(j) for i, j in offsets) == 2
    assert plan.field_extents["in_field"] == ((-2, 2), (-2, 2))  # unchanged by the rewrite
    assert all(n.startswith(("lap_field__v", "res__v", "flx_field__v", "fly_field__v")) for n in plan.locals)


def test_parallel_computations_with_vertical_dependency_run_one_after_the_other(programs):
    """One column kernel; each PARALLEL computation is a K loop of its own, completed before the next starts."""
    (stage,) = programs["parallel_k_dependency"].plan.stages
    assert stage.mapping == "column" and len(stage.nests) == 3


def test_loop_invariant_temporaries_leave_the_sequential_block(programs):
    """A FORWARD block that reads, at a horizontal offset, a temporary computed from fields the block does not write:
    the temporary becomes a PARALLEL stage of its own and the sweep stays a thread per column."""
    stages = programs["cross_column_recurrence"].plan.stages
    assert all(st.plane is None for st in stages)
    assert [(st.mapping, [(n.order.value, [s.target.name for s in n.stmts]) for n in st.nests]) for st in stages][:2] == [
        ("ijk", [("parallel", ["t"])]),
        ("column", [("forward", ["a"]), ("backward", ["c"]), ("parallel", ["u"])])]


def test_sequential_blocks_with_cross_column_dependencies_run_plane_by_plane(programs):
    """... and when the temporary takes part in the sweep: thread-per-point stages (cut between producer and
    offset reader) that the host launches once per K level."""
    stages = programs["plane_recurrence"].plan.stages
    assert [st.mapping for st in stages] == ["column", "ijk", "ijk"]
    assert [st.plane and (st.plane[0], st.plane[1].value) for st in stages] == [None, (1, "forward"), (1, "forward")]
    assert [[s.target.name for n in st.nests for s in n.stmts] for st in stages[1:]] == [["s", "t"], ["out"]]
    assert [k.plane is not None for k in programs["plane_recurrence"].kernels] == [False, True, True]


def test_unsupported_shapes_are_rejected_loudly():
    def neighbours_inside_a_while(a: "Field[np.float64]", b: "Field[np.float64]"):  # noqa: F821
        with computation(PARALLEL), interval(...):  # noqa: F821
            t = a
            n = 0
            while n < 2:
                u = t[1, 0, 0]
                t = u + b
                n = n + 1
            a = t

    with pytest.raises(NotImplementedError, match="written and read at a horizontal offset inside one statement"):
        gtscript.stencil(backend="hip:mi300", definition=neighbours_inside_a_while)


@pytest.mark.parametrize("name", sorted(zoo.ZOO))
def test_rewritten_ir_is_value_equivalent(programs, name):
    """Run the ORIGINAL and the REWRITTEN (inlined, SSA) IR through the numpy oracle: same bits."""
    _, _, scalars, _ = zoo.ZOO[name]
    ref = _build(name, "numpy")
    domain = (6, 5, 7)
    arrays, origins = zoo.make_inputs(ref, domain)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, **scalars, origin=origins, domain=domain)
    got = {k: v.copy() for k, v in arrays.items()}
    rewritten = programs[name].plan.stencil
    oracle_backend.run_stencil(rewritten, analysis.compute_extents(rewritten), domain, origins, got, scalars)
    for k in arrays:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"{name}: field {k}")


@pytest.mark.parametrize("name", sorted(zoo.ZOO))
def test_generated_source_compiles_for_gfx950(programs, name):
    prog = programs[name]
    for options in ([], ["-DGT4MI_UNIT_I_STRIDE=1", "-DGT4MI_NO_ALIAS=1"]):
        code = _lib.rtc_compile(prog.source, f"{name}.hip", options)
        assert code[:4] == b"\x7fELF"
        for kern in prog.kernels:
            assert kern.name.encode() in code
    # the argument block mirrors `struct gt_args`: pointers + 3 strides per array, scalars, 3 extents
    n_arrays = len(prog.plan.api_fields) + len(prog.plan.scratch)
    n_data = sum(len(d.data_dims) for d in (*prog.plan.api_fields, *prog.plan.stencil.temporaries)
                 if d.name in prog.plan.scratch or d in prog.plan.api_fields)
    assert len(prog.args_struct._fields_) == 4 * n_arrays + n_data + len(prog.plan.params) + 6  # dI, dJ, dK, k_lo, k_hi, lead


def test_horizontal_stages_get_a_16_byte_lane_twin(programs):
    """Thread-per-point stages that only read arrays they do not write also exist as `<name>_vec`: a lane
    owns 2 fp64 (4 fp32) consecutive I points and several J rows; neighbours come from DPP lane shifts."""
    lap = programs["laplacian"]
    # a light stage (one statement, one 8-byte array, offsets within one row / column): 8 rows per lane, XCD runs of 4
    assert [(k.vec, k.vec_rows, k.vec_fields) for k in lap.kernels] == [(2, 8, ("out", "inp"))]
    assert "gt4mi_laplacian_stage0_vec" in lap.source and "gt_shift<double, true>" in lap.source
    assert "gt_tile(4u, gt_bx, gt_by, gt_bz);" in lap.source[lap.source.index("gt4mi_laplacian_stage0_vec"):]
    # everything heavier keeps 4 rows and the plain tile order
    hd = programs["horizontal_diffusion"]
    vec_part = hd.source[hd.source.index("gt4mi_horizontal_diffusion_stage0_vec("):hd.source.index("gt4mi_horizontal_diffusion_stage0_vecs(")]
    assert [(k.vec, k.vec_rows) for k in hd.kernels] == [(2, 4)] and "gt_tile(4u" not in vec_part
    assert [k.vec for k in programs["horizontal_diffusion_f32"].kernels] == [4]
    assert [k.vec for k in programs["mixed_precision"].kernels] == [2]  # widest element decides
    # not vectorised: sequential stages, stages that read what they write, arrays without an I axis as target
    assert [k.vec for k in programs["tridiagonal_solver"].kernels] == [0]
    assert [k.vec for k in programs["two_stage_written_input"].kernels] == [0, 2]
    assert [k.vec for k in programs["column_sum_then_gradient"].kernels] == [0, 2]


def test_stages_with_inlined_temporaries_get_a_strip_kernel_that_shares_them(programs):
    """`<name>_vecs`: every inlined temporary is computed once per point, at the lane's own columns, and read at an I
    offset through a DPP shift of the neighbouring lane's value; waves overlap by a halo lane on each side."""
    import re

    for name, vec in (("horizontal_diffusion", 2), ("horizontal_diffusion_f32", 4)):
        prog = programs[name]
        (kern,) = prog.kernels
        rows = 8 if vec == 4 else 4
        assert (kern.vec, kern.shared_halo, kern.shared_rows) == (vec, 1, rows), name  # inputs reach 2 columns: one halo lane
        src = prog.source[prog.source.index(f"gt4mi_{name}_stage0_vecs("):]
        assert f"const gt_i64 i0 = (wave_x * 62 - 1 + lane) * {vec} - a.lead;" in src and "const bool out_lane = lane >= 1 && lane < 63;" in src
        # lap on rows -1 .. rows (rows + 2 of them, x vec components), each exactly once; the recomputing kernel derives
        # it per consumer
        laps = re.findall(r"const double (t_lap\w*?__v0_[mp]\d_\d) = ", src)
        assert len(laps) == len(set(laps)) == (rows + 2) * vec, laps
        assert "gt_shift<double, true, true>(t_lap" in src or "gt_shift<double, false, true>(t_lap" in src  # a temporary crosses lanes
    # a chain that reaches 3 columns: no recomputing strip kernel (its fix-ups reach one lane), but the sharing one with two
    # halo lanes; two interval blocks that use the same temporary names: one stage, no scratch
    hyper = programs["hyperdiffusion_6th"]
    assert [(k.vec, k.shared_vec, k.shared_halo) for k in hyper.kernels] == [(0, 2, 2)] and not hyper.plan.scratch
    assert len(hyper.plan.stages) == 1 and len(hyper.plan.stages[0].nests) == 2
    assert "const gt_i64 i0 = (wave_x * 60 - 2 + lane) * 2 - a.lead;" in hyper.source
    # the limiter written as if / else blocks: the conditionally assigned fluxes are selects, one stage, no scratch
    hd_if = programs["horizontal_diffusion_if"]
    assert len(hd_if.plan.stages) == 1 and not hd_if.plan.scratch and [(k.vec, k.shared_halo) for k in hd_if.kernels] == [(2, 1)]
    # nothing to share: no temporaries (Laplacian), or no horizontal offsets on them
    assert [k.shared_halo for k in programs["laplacian"].kernels] == [0]
    assert all(k.shared_halo == 0 for k in programs["vertical_advection_dycore"].kernels)


def test_two_sweep_column_stages_get_a_ladder_of_cached_variants(programs):
    """FORWARD-then-BACKWARD column stages also exist as `<name>_tc<n>`: the top n + LDS levels of the fields the second
    sweep reads back stay on chip (stage_planner.TopCache).  What keeps a cached level at the cost of its own registers
    is pinned here, because no parity test notices when it is lost (the kernels only get slower, or spill and fall back):
    the register ladder, the pin after every register level, the opaque bases of the second sweep, streaming stores
    and the rolling prefetch."""
    import re

    vadv = programs["vertical_advection_dycore"]
    (kern,) = vadv.kernels
    assert kern.top_cache == tuple((n, 40, n + 42) for n in (112, 104, 80, 56, 32, 16))  # 2 fp64 fields, 448 dwords, 160 KB LDS
    assert [k.top_cache for k in programs["tridiagonal_solver"].kernels] == [kern.top_cache]
    assert programs["two_sweep_three_carried"].kernels[0].top_cache == tuple((n, 32, n + 35) for n in (88, 80, 56, 32, 16))
    assert all(k.top_cache is None for k in programs["column_sum_then_gradient"].kernels)  # one sweep: nothing to keep
    src = vadv.source
    tc = src[src.index("gt4mi_vertical_advection_dycore_stage0_tc104("):src.index("gt4mi_vertical_advection_dycore_stage0_tc80(")]
    # every register level of the first sweep is pinned where it is computed (levels 0 .. 102 write ccol and dcol, the last only dcol)
    assert len(re.findall(r'asm volatile\("" :: "v"\(tc_ccol_\d+\), "v"\(tc_dcol_\d+\)\);', tc)) == 103
    assert 'asm volatile("" :: "v"(tc_dcol_103));' in tc
    # the second sweep addresses its arrays through bases the optimiser cannot relate to the first sweep's
    assert 'asm volatile("" : "+s"(tc_zero));' in tc and "const auto tc_b_u_pos = b_u_pos + tc_zero;" in tc
    second = tc[tc.index("tc_zero"):]
    # its results are never read again by the kernel: streaming stores; spilled ccol / dcol levels are ordinary stores
    assert re.search(r"__builtin_nontemporal_store\(.*&b_utens_stage\[", second)
    first = tc[:tc.index("tc_zero")]
    assert re.search(r"\bb_ccol\[.*\] = ", first) and not re.search(r"__builtin_nontemporal_store\(.*&b_ccol\[", first)
    # rolling prefetch over the register levels: loads are named q<n>_<field>, one barrier per level, and a value the
    # sweep already holds (wcon of the level above) is not loaded a second time
    reg = first[first.rindex("gt_min(k1, a.dK - 104)"):]
    assert len(re.findall(r"const double q\d+_wcon = ", reg)) <= 2 * 105 + 2
    assert reg.count("__builtin_amdgcn_sched_barrier(0);") >= 104


def test_compiler_errors_surface_with_the_log():
    with pytest.raises(_lib.NativeError, match="expected ';'"):
        _lib.rtc_compile('extern "C" __global__ void k(double* a) { a[0] = 1.0 }')


def test_code_objects_are_cached_on_disk(programs, tmp_path, monkeypatch):
    """Counterpart of the reference's `.gt_cache`: one gfx950 code object per (source, options)."""
    from gt4py_amd.cartesian.backend import hip_generic

    monkeypatch.setenv("GT4PY_AMD_CACHE_DIR", str(tmp_path))
    prog = programs["copy_stencil"]
    calls = []
    real = _lib.rtc_compile
    monkeypatch.setattr(_lib, "rtc_compile", lambda *a, **k: (calls.append(a[1]), real(*a, **k))[1])
    first = hip_generic._compile_cached(prog.source, "copy.hip", ["-DGT4MI_UNIT_I_STRIDE=1"])
    again = hip_generic._compile_cached(prog.source, "copy.hip", ["-DGT4MI_UNIT_I_STRIDE=1"])
    other = hip_generic._compile_cached(prog.source, "copy.hip", [])
    assert first == again and first[:4] == b"\x7fELF" and len(calls) == 2  # the second call came from disk
    assert len(list(tmp_path.glob("*.hsaco"))) == 2 and other[:4] == b"\x7fELF"
    monkeypatch.setenv("GT4PY_AMD_CACHE_DIR", "")
    hip_generic._compile_cached(prog.source, "copy.hip", [])
    assert len(calls) == 3  # caching switched off: compiled again


# ---- the planner's rewriting passes on programs written for them (tests/planner_cases/programs.py) -------------------
def _planned(defn):
    obj = gtscript.stencil(backend="hip:mi300", definition=defn)
    return obj, type(obj)._gt_program_.plan


def _statements(plan):
    return [(s.target.name, s.mask is not None) for c in plan.stencil.computations for b in c.blocks for s in b.body]


def _rewrite_matches_the_original(defn, domain=(7, 6, 4)):
    ref = gtscript.stencil(backend="numpy", definition=defn)
    _, plan = _planned(defn)
    arrays, origins = zoo.make_inputs(ref, domain, 3)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, origin=origins, domain=domain)
    got = {k: v.copy() for k, v in arrays.items()}
    oracle_backend.run_stencil(plan.stencil, analysis.compute_extents(plan.stencil), domain, origins, got, {})
    for k in arrays:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=k)


def test_temporaries_that_blocks_use_each_for_itself_are_split_per_block():
    from planner_cases import programs as P

    _, plan = _planned(P.boundary_and_interior)
    names = [n for n, _ in _statements(plan)]
    assert len(plan.stages) == 1 and not plan.scratch  # one kernel, nothing through memory
    assert any(n.startswith("lap__b0_0") for n in names) and any(n.startswith("lap__b0_1") for n in names), names
    _rewrite_matches_the_original(P.boundary_and_interior)
    # a value that does flow from one block into another stays one temporary, in memory, with a stage cut
    _, plan = _planned(P.value_crosses_blocks)
    assert sorted(plan.scratch) == ["t"] and len(plan.stages) == 2
    _rewrite_matches_the_original(P.value_crosses_blocks)


def test_if_else_pairs_become_one_conditional_expression_unless_something_interferes():
    from planner_cases import programs as P

    _, plan = _planned(P.if_else_pair)
    # `out` assigned by both branches: one plain statement; `other` by one branch only: still conditional
    assert _statements(plan) == [("mask_0", False), ("other", True), ("out", False)]
    _rewrite_matches_the_original(P.if_else_pair)
    _, plan = _planned(P.if_else_with_interference)
    # the else branch of `out` reads the `t` the if branch assigned in between, and vice versa: nothing may be merged
    assert _statements(plan) == [("t", False), ("mask_0", False), ("out", True), ("t", True), ("out", True), ("t", True)]
    _rewrite_matches_the_original(P.if_else_with_interference)
    # a statement between two conditional assignments rewrites what the condition reads
    _rewrite_matches_the_original(P.condition_input_rewritten_between_branches)


def test_top_of_column_cache_only_takes_fields_every_level_of_the_first_sweep_assigns():
    """a field the FORWARD sweep assigns at the boundary level only and the BACKWARD sweep reads on
    every level keeps the caller's values above that level -- the second sweep must read memory, not an on-chip cache that
    nobody filled."""
    from planner_cases import programs as P

    _, plan = _planned(P.boundary_only_write_read_back)
    assert [tc.names for tc in plan.top_cache.values()] == [("s",)]
    _rewrite_matches_the_original(P.boundary_only_write_read_back, domain=(5, 4, 9))



def test_elements_disjoint_never_claims_more_than_brute_force():
    """hip_generic.elements_disjoint (and its C++ twin in csrc/common.hip.h) against the address sets of random slices of
    one parent array: a claim of disjointness must always be true; the interleaved and the halves cases must be proven."""
    from gt4py_amd.cartesian.backend.hip_generic import elements_disjoint

    rng = np.random.default_rng(0)

    def addresses(v):
        idx = np.indices(v.shape).reshape(v.ndim, -1)
        return set((v.__array_interface__["data"][0] + (idx * np.array(v.strides)[:, None]).sum(0)).tolist())

    claims = 0
    for _ in range(400):
        shape = rng.integers(2, 6, size=4)
        parent = np.zeros(shape, order=str(rng.choice(["C", "F"])))

        def view():
            sl = []
            for n in shape[:3]:
                a = int(rng.integers(0, n))
                sl.append(slice(a, int(rng.integers(a + 1, n + 1))))
            return parent[tuple(sl) + (int(rng.integers(0, shape[3])),)]

        a, b = view(), view()
        got = elements_disjoint(a.__array_interface__["data"][0], a.shape, b.__array_interface__["data"][0], b.shape, a.strides, 8)
        if got:
            claims += 1
            assert not (addresses(a) & addresses(b))
    assert claims > 100
    vel = np.zeros((6, 5, 4, 2))
    p0, p1 = (vel[..., n].__array_interface__["data"][0] for n in (0, 1))
    assert elements_disjoint(p0, (6, 5, 4), p1, (6, 5, 4), vel[..., 0].strides, 8)
    assert not elements_disjoint(p0, (6, 5, 4), p0 + vel.strides[1], (6, 5, 4), vel[..., 0].strides, 8)  # shifted by one row


def test_column_kernels_load_read_once_streams_nontemporally():
    """Round 5 (profiles/r5_nt_loads_column_kernels.txt): in a column kernel the fields read at no horizontal offset that only ONE
    sweep reads from memory are loaded with `__builtin_nontemporal_load` (+7 % on the vertical advection); `wcon`, which the
    neighbouring lane reads too, never is, and `u_pos`, which both sweeps read, only at its last use; mode 0 emits plain loads."""
    import re

    import stencil_zoo as zoo
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_codegen

    defn, ext, _, opts = zoo.ZOO["vertical_advection_dycore"]
    saved = hip_codegen.TUNING["column_nt_loads"]
    try:
        found = {}
        for mode in (0, 3, 5):
            hip_codegen.TUNING["column_nt_loads"] = mode
            st = gtscript.stencil(backend="hip:mi300", definition=defn, externals=ext, rebuild=True, name=f"vadv_nt_mode_{mode}", **opts)
            src = type(st)._gt_program_.source
            found[mode] = set(re.findall(r"__builtin_nontemporal_load\(&\w*?b2?_(\w+?)\[", src))
    finally:
        hip_codegen.TUNING["column_nt_loads"] = saved
    assert found[0] == set()
    assert {"u_stage", "utens", "utens_stage"} <= found[3] and "wcon" not in found[3] and "u_pos" not in found[3]
    assert found[5] == found[3] | {"u_pos"}


def test_strip_kernels_load_arrays_read_at_their_own_point_only_nontemporally():
    """... and in the 16-byte-lane kernels of horizontal stages: horizontal diffusion's `coeff` (read exactly once), never `in_field`."""
    import stencil_zoo as zoo
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.backend import hip_codegen

    defn, ext, _, opts = zoo.ZOO["horizontal_diffusion"]
    st = gtscript.stencil(backend="hip:mi300", definition=defn, rebuild=True, name="hdiff_nt_strip", **opts)
    program = type(st)._gt_program_
    stage = program.plan.stages[0]
    em = hip_codegen._Emitter(program.plan)
    assert hip_codegen._read_once_fields(em, stage) == {"coeff"}
    assert "__builtin_nontemporal_load(reinterpret_cast<const gt_vec<double, 2>*>" in program.source
