"""GPU: the generic executor (generated HIP, compiled by hiprtc on the box) vs the numpy oracle.

Every stencil of tests/stencil_zoo.py is built twice -- backend "numpy" (oracle/numpy_backend.py, the
restatement of the reference's numpy backend) and backend "hip:mi300" -- and run on the same seeded
inputs; every field (outputs AND inputs, so stray writes show) must match bit for bit, NaNs included.
The shapes that the hand-written kernel library also covers are additionally compared with it.
"""

import numpy as np
import pytest

import stencil_zoo as zoo

pytestmark = pytest.mark.gpu

DOMAINS = [(1, 1, 3), (3, 5, 4), (17, 33, 5), (64, 64, 8), (65, 63, 7), (130, 9, 40)]


def _run_pair(name, domain, seed=1337, layout_backend="hip:mi300"):
    import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    defn, externals, scalars, opts = zoo.ZOO[name]
    ref = gtscript.stencil(backend="numpy", definition=defn, externals=externals)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, **opts)
    assert hasattr(type(hip), "_gt_program_"), "must run through the generic executor"
    assert ref.field_info == hip.field_info and ref.domain_info == hip.domain_info
    if domain[2] < ref.domain_info.min_sequential_axis_size:
        pytest.skip(f"K size {domain[2]} below the stencil's minimum")
    arrays, origins = zoo.make_inputs(ref, domain, seed)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, **scalars, origin=origins, domain=domain)
    dev = {}
    for k, v in arrays.items():
        dims = hip.field_info[k].axes
        dev[k] = gt_storage.from_array(v, dtype=v.dtype, backend=layout_backend, aligned_index=origins[k],
                                       dimensions=dims)
    hip(**dev, **scalars, origin=origins, domain=domain)
    return expect, {k: d.get() if hasattr(d, "get") else np.asarray(d) for k, d in dev.items()}, hip


@pytest.mark.parametrize("name", sorted(zoo.ZOO))
@pytest.mark.parametrize("domain", DOMAINS)
def test_generated_kernels_match_the_oracle(name, domain):
    expect, got, _ = _run_pair(name, domain)
    for k in expect:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"{name} {domain}: field {k}")


@pytest.mark.parametrize("name", ["horizontal_diffusion", "vertical_advection_dycore", "column_sum_then_gradient"])
def test_strided_layout_variant(name):
    """K-contiguous (numpy-layout) device arrays: the non-unit-I-stride variant is compiled and agrees."""
    import torch
    import oracle.numpy_backend  # noqa: F401
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.storage import as_device_array

    defn, externals, scalars, opts = zoo.ZOO[name]
    ref = gtscript.stencil(backend="numpy", definition=defn, externals=externals)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, **opts)
    domain = (19, 21, 6)
    arrays, origins = zoo.make_inputs(ref, domain, seed=7)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, **scalars, origin=origins, domain=domain)
    dev = {k: torch.from_numpy(v).cuda() for k, v in arrays.items()}  # C order: K contiguous
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # non-optimal layout warning, as in the reference
        hip(**dev, **scalars, origin=origins, domain=domain)
    for k in expect:
        np.testing.assert_array_equal(dev[k].cpu().numpy(), expect[k], err_msg=f"{name}: field {k}")
    assert (False, True) in type(hip)._gt_variants_


@pytest.mark.parametrize("name", ["laplacian", "horizontal_diffusion", "horizontal_diffusion_f32", "if_with_offsets",
                                  "hyperdiffusion_6th", "horizontal_diffusion_if"])
def test_misaligned_origin_falls_back_to_one_point_per_thread(name):
    """The 16-byte-lane kernels need every lane's vector naturally aligned.  An origin on an odd column (the array is
    aligned on column 0 instead) either starts the lanes of the `_vec` strip kernel that far before the domain (round 3:
    `a.lead`; single-stage stencils whose arrays are all equally misaligned; the `_vecs` kernels with temporaries shared
    between lanes do the same) or selects the scalar twin -- and all of them must agree with the oracle."""
    import oracle.numpy_backend  # noqa: F401
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    defn, externals, scalars, opts = zoo.ZOO[name]
    ref = gtscript.stencil(backend="numpy", definition=defn, externals=externals)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, **opts)
    for shift, domain in ((1, (37, 11, 5)), (3, (37, 11, 5)), (1, (130, 9, 3)), (2, (251, 10, 2))):  # one wave and several
        arrays, origins = zoo.make_inputs(ref, domain, seed=5)
        arrays = {k: np.pad(v, ((shift, 0), (0, 0), (0, 0))) for k, v in arrays.items()}
        origins = {k: (o[0] + shift, o[1], o[2]) for k, o in origins.items()}
        expect = {k: v.copy() for k, v in arrays.items()}
        ref(**expect, **scalars, origin=origins, domain=domain)
        dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300",
                                        aligned_index=(origins[k][0] - shift, origins[k][1], 0)) for k, v in arrays.items()}
        hip(**dev, **scalars, origin=origins, domain=domain)
        for k in expect:
            np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"{name} shift {shift}: field {k}")


def test_aliased_arguments_use_the_aliasing_safe_variant():
    """copy(a, a) hands the same buffer twice: the no-alias variant must not be chosen."""
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    hip = gtscript.stencil(backend="hip:mi300", definition=zoo.copy_stencil)
    a = gt_storage.from_array(np.arange(24.0).reshape(2, 3, 4), backend="hip:mi300")
    hip(a, a)
    assert np.array_equal(a.get(), np.arange(24.0).reshape(2, 3, 4))
    assert (True, False) in type(hip)._gt_variants_  # (unit I stride, arrays disjoint)
    b = gt_storage.zeros((2, 3, 4), backend="hip:mi300")
    hip(a, b)
    assert np.array_equal(b.get(), a.get()) and (True, True) in type(hip)._gt_variants_


@pytest.mark.parametrize("name,family", [("horizontal_diffusion", "hdiff"), ("laplacian", "lap5"),
                                         ("tridiagonal_solver", "tridiag"), ("horizontal_diffusion_f32", "hdiff")])
def test_generic_and_hand_written_kernels_agree(name, family):
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    defn, externals, scalars, _ = zoo.ZOO[name]
    lib_obj = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals)
    assert type(lib_obj)._gt_binding_.family == family
    domain = (70, 37, 9)
    expect, got, _ = _run_pair(name, domain, seed=11)
    arrays, origins = zoo.make_inputs(lib_obj, domain, seed=11)
    dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k])
           for k, v in arrays.items()}
    lib_obj(**dev, **scalars, origin=origins, domain=domain)
    for k in expect:
        np.testing.assert_array_equal(dev[k].get(), got[k], err_msg=f"{name}: field {k}")


def test_exec_info_and_frozen_call_on_the_generic_path():
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    defn, externals, scalars, opts = zoo.ZOO["vertical_advection_dycore"]
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals)
    domain = (32, 16, 12)
    arrays, origins = zoo.make_inputs(hip, domain, seed=3)
    dev = {k: gt_storage.from_array(v, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
    info = {}
    hip(**dev, **scalars, origin=origins, domain=domain, exec_info=info)
    assert info["run_cpp_end_time"] >= info["run_cpp_start_time"] > 0
    first = dev["utens_stage"].get().copy()
    frozen = hip.freeze(origin=origins, domain=domain)
    dev2 = {k: gt_storage.from_array(v, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
    frozen(**dev2, **scalars)
    assert np.array_equal(dev2["utens_stage"].get(), first)


# ---- arguments that overlap in memory, calls on several streams -------------------------------------------------
def _alias_definitions():
    from gt4py_amd.cartesian.gtscript import FORWARD, PARALLEL, Field, computation, interval  # noqa: F401

    def pointwise(a: Field[np.float64], w: Field[np.float64], out: Field[np.float64]):
        with computation(PARALLEL), interval(...):
            t = a * 2.0 + w
            out = t - a if w > 0.0 else t + a

    def shifted(a: Field[np.float64], out: Field[np.float64]):
        with computation(PARALLEL), interval(...):
            out = a[1, 0, 0] - a

    def two_outputs(a: Field[np.float64], b: Field[np.float64], c: Field[np.float64]):
        with computation(PARALLEL), interval(...):
            b = a + 1.0
            c = a - 1.0

    return pointwise, shifted, two_outputs


def test_aliased_arguments_follow_the_reference_or_raise():
    """npir_codegen.py:205-210: the numpy backend evaluates a right-hand side before it assigns, so passing one array
    for two fields is well defined there.  The generic executor reproduces it where the order of evaluation cannot
    matter (a written field sharing its elements with a read-only one, all accesses at zero offset) and raises
    otherwise -- compared with the oracle called on the SAME aliased arrays."""
    import oracle.numpy_backend  # noqa: F401
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    pointwise, shifted, two_outputs = _alias_definitions()
    rng = np.random.default_rng(9)
    a, w = rng.uniform(-1, 1, (40, 9, 5)), rng.uniform(-1, 1, (40, 9, 5))
    ref = gtscript.stencil(backend="numpy", definition=pointwise)
    hip = gtscript.stencil(backend="hip:mi300", definition=pointwise)
    for shared in ("a", "w"):
        host = {"a": a.copy(), "w": w.copy()}
        host["out"] = host[shared]
        ref(host["a"], host["w"], host["out"])
        dev = {"a": gt_storage.from_array(a, backend="hip:mi300"), "w": gt_storage.from_array(w, backend="hip:mi300")}
        dev["out"] = dev[shared]
        hip(dev["a"], dev["w"], dev["out"])
        for n in ("a", "w"):
            np.testing.assert_array_equal(dev[n].get(), host[n], err_msg=f"out is {shared}: field {n}")
    # read-only arguments may be the same array
    dev_a = gt_storage.from_array(a, backend="hip:mi300")
    dev_o = gt_storage.zeros(a.shape, backend="hip:mi300")
    want = np.zeros_like(a)
    ref(a, a, want)
    hip(dev_a, dev_a, dev_o)
    np.testing.assert_array_equal(dev_o.get(), want)
    # a written field that is read at an offset through its alias: refused, untouched
    sh = gtscript.stencil(backend="hip:mi300", definition=shifted)
    dev_a = gt_storage.from_array(a, backend="hip:mi300")
    with pytest.raises(ValueError, match="same array"):
        sh(dev_a, dev_a, origin=(0, 0, 0), domain=(39, 9, 5))
    # ... or through a shifted view of the same buffer
    pw = gtscript.stencil(backend="hip:mi300", definition=pointwise)
    with pytest.raises(ValueError, match="overlap in memory"):
        pw(dev_a, dev_o, dev_a, origin={"a": (0, 0, 0), "w": (0, 0, 0), "out": (1, 0, 0)}, domain=(39, 9, 5))
    # two outputs in one array: the reference keeps the second assignment; refused here
    two = gtscript.stencil(backend="hip:mi300", definition=two_outputs)
    with pytest.raises(ValueError, match="writes both"):
        two(dev_a, dev_o, dev_o)
    np.testing.assert_array_equal(dev_a.get(), a)


def test_calls_on_two_streams_do_not_share_scratch():
    """A stencil with temporaries in scratch memory, launched on two HIP streams at once with different inputs: each
    stream has its own scratch buffer, so both results equal the oracle's."""
    import torch

    import oracle.numpy_backend  # noqa: F401
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    defn, externals, scalars, opts = zoo.ZOO["two_stage_written_input"]
    ref = gtscript.stencil(backend="numpy", definition=defn, externals=externals)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, device_sync=False, **opts)
    assert type(hip)._gt_program_.plan.scratch, "the case must use scratch memory"
    domain = (256, 192, 24)
    for round_ in range(3):
        cases = []
        for seed in (10 * round_ + 1, 10 * round_ + 2):
            arrays, origins = zoo.make_inputs(ref, domain, seed)
            expect = {k: v.copy() for k, v in arrays.items()}
            ref(**expect, **scalars, origin=origins, domain=domain)
            dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k])
                   for k, v in arrays.items()}
            cases.append((expect, dev, origins))
        streams = [torch.cuda.Stream(), torch.cuda.Stream()] if round_ == 0 else streams
        torch.cuda.synchronize()
        for (expect, dev, origins), st in zip(cases, streams):
            with torch.cuda.stream(st):
                hip(**dev, **scalars, origin=origins, domain=domain)
        torch.cuda.synchronize()
        for expect, dev, _ in cases:
            for k in expect:
                np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"round {round_}: field {k}")
    assert len({k[0] for k in type(hip)._gt_scratch_}) == 2, "one scratch buffer per stream"


# ---- two-sweep column stages with the top of the column kept on chip (the `_tc` kernel variant) -------------------
TWO_SWEEP = ["tridiagonal_solver", "vertical_advection_dycore", "two_sweep_three_carried"]


def _run_pair_rebuilt(name, domain, seed=4242):
    """like _run_pair, but the hip:mi300 stencil is rebuilt (the generator's tuning knobs are not part of the cache key)"""
    import oracle.numpy_backend  # noqa: F401
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript

    defn, externals, scalars, opts = zoo.ZOO[name]
    ref = gtscript.stencil(backend="numpy", definition=defn, externals=externals)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, rebuild=True, **opts)
    arrays, origins = zoo.make_inputs(ref, domain, seed)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, **scalars, origin=origins, domain=domain)
    dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k],
                                    dimensions=hip.field_info[k].axes) for k, v in arrays.items()}
    hip(**dev, **scalars, origin=origins, domain=domain)
    return expect, {k: d.get() for k, d in dev.items()}, hip


@pytest.mark.parametrize("name", TWO_SWEEP)
@pytest.mark.parametrize("domain", [(70, 5, 100), (130, 3, 161), (64, 8, 58), (64, 8, 59), (33, 2, 57), (66, 3, 73), (66, 3, 74),
                                    (66, 3, 75), (66, 3, 80), (66, 3, 97), (66, 3, 98), (66, 3, 99), (66, 3, 121), (66, 3, 122),
                                    (66, 3, 123), (66, 3, 137), (66, 3, 145), (66, 3, 146), (66, 3, 147), (66, 3, 153),
                                    (66, 3, 154), (66, 3, 155), (64, 2, 200)])
def test_top_of_column_cache_default_depths(name, domain):
    """Deep domains take a `_tc<n>` kernel (registers + LDS hold the top levels between the sweeps) -- the deepest one
    the domain has room for --, shallower ones the plain kernel; every one must reproduce the oracle bit for bit on
    every field.  The domains sit on both sides of every variant's smallest K."""
    from gt4py_amd.cartesian.backend import hip_codegen

    expect, got, hip = _run_pair_rebuilt(name, domain)
    kern = type(hip)._gt_program_.kernels[0]
    # register levels: 448 dwords per lane / dwords per cached level, rounded down to a multiple of 8; 8 fewer; then every 24
    # levels, and 16; LDS levels = 160 KB / (bytes per column and level x 256 threads); smallest domain = both + margin + 1
    want = {"two_sweep_three_carried": tuple((n, 32, n + 32 + 3) for n in (88, 80, 56, 32, 16))}.get(
        name, tuple((n, 40, n + 40 + 2) for n in (112, 104, 80, 56, 32, 16)))
    assert kern.top_cache == want, hip_codegen.TUNING["top_cache"]
    for k in expect:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"{name} {domain}: field {k}")


@pytest.mark.parametrize("name", TWO_SWEEP)
@pytest.mark.parametrize("depths", [(3, 5), (1, 0), (0, 4), (5, 1), (8, 8)])
@pytest.mark.parametrize("levels", [8, 9, 10, 11, 12, 13, 18, 19, 40])
def test_top_of_column_cache_every_range_boundary(name, depths, levels):
    """Shallow caches (a few register levels, a few LDS levels) on domains around the smallest one the variant accepts:
    every combination of empty / one-level / several-level memory, LDS and register ranges in both sweeps."""
    from gt4py_amd.cartesian.backend import hip_codegen

    saved = hip_codegen.TUNING["top_cache"]
    items = {"two_sweep_three_carried": 8 + 4 + 8}.get(name, 2 * 8)  # bytes per column and level of the cached fields
    per_level = items * 256  # 256 threads per workgroup
    hip_codegen.TUNING["top_cache"] = (depths[0], depths[1] * per_level)
    try:
        expect, got, hip = _run_pair_rebuilt(name, (66, 5, levels), seed=levels)
        kern = type(hip)._gt_program_.kernels[0]
        margin = {"two_sweep_three_carried": 2}.get(name, 1)
        assert kern.top_cache == ((depths[0], depths[1], depths[0] + depths[1] + margin + 1),)
    finally:
        hip_codegen.TUNING["top_cache"] = saved
        # leave no class built with the shallow depths in the cache
        defn, externals, _, opts = zoo.ZOO[name]
        from gt4py_amd.cartesian import gtscript

        gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, rebuild=True, **opts)
    for k in expect:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"{name} depths {depths} K={levels}: field {k}")


def test_top_of_column_cache_is_what_runs_and_does_not_spill():
    """The deep variant is really the one launched for K = 160, uses no scratch memory, and a stencil object that was
    first called on a shallow domain switches to it when the domain grows."""
    import ctypes

    from gt4py_amd import _lib

    expect, got, hip = _run_pair_rebuilt("vertical_advection_dycore", (64, 4, 20))
    for k in expect:
        np.testing.assert_array_equal(got[k], expect[k])
    variant = next(iter(type(hip)._gt_variants_.values()))
    tfns = variant.tc_functions[0]
    # (the 112-level variant, smallest K 154, needs a few registers more than a lane has with the rolling prefetch: refused)
    assert [min_k for _, min_k in tfns] == [146, 122, 98, 74, 58], "a `_tc<n>` kernel was refused (spills?)"
    for tfn, _ in tfns:
        regs, scratch, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check("gt4mi_function_info", _lib.load().gt4mi_function_info(tfn, ctypes.byref(regs), ctypes.byref(scratch), ctypes.byref(lds)))
        assert scratch.value == 0 and lds.value == 160 * 1024 and regs.value <= 512
    expect, got, hip2 = _run_pair_rebuilt("vertical_advection_dycore", (64, 4, 160))
    for k in expect:
        np.testing.assert_array_equal(got[k], expect[k])


# ---- strip kernels whose inlined temporaries are shared between lanes (the `_vecs` kernel variant) -------------------
@pytest.mark.parametrize("name", ["horizontal_diffusion", "horizontal_diffusion_f32"])
@pytest.mark.parametrize("domain", [(1, 5, 2), (2, 6, 2), (123, 5, 3), (124, 10, 2), (125, 11, 2), (247, 9, 2), (248, 15, 3), (249, 4, 2),
                                    (300, 23, 3), (496, 5, 1), (497, 7, 2), (1000, 26, 2)])
def test_shared_temporaries_strip_kernel_at_every_edge(name, domain):
    """Waves of the `_vecs` kernel cover 62 lanes x 2 (fp64) or 4 (fp32) columns and overlap by a halo lane on each side;
    strips are 4 (fp64) or 8 (fp32) rows.  Domains on both sides of one, two and four waves in I, with whole and partial strips in J: the
    halo lanes at the domain edge read only what the arrays hold, the last wave stores only inside the domain, the rows
    that do not fill a strip take the point-by-point path -- all bit-identical to the oracle."""
    expect, got, hip = _run_pair(name, domain, seed=sum(domain))
    kern = type(hip)._gt_program_.kernels[0]
    assert kern.shared_halo == 1 and kern.shared_rows == (8 if name.endswith("f32") else 4)
    # (the class may hold other flavours from earlier calls -- strided or aliased arguments --, which have no `_vecs` twin)
    assert any(v.shared_functions[0] is not None for v in type(hip)._gt_variants_.values()), \
        "the `_vecs` kernel must be what a call with aligned, disjoint storages launches"
    for k in expect:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"{name} {domain}: field {k}")


@pytest.mark.parametrize("inline_masked", [False, True])
def test_conditionally_assigned_temporaries_across_a_stage_cut(inline_masked, monkeypatch):
    """Two planner bugs the shared-temporaries fuzzer found (seed 793 and the `elif` chain of the reference's
    set_inner_as_kord), pinned on a small program: (1) `t = a` and, further down, `if c: t = b` with a stage cut in
    between -- the old value shows through where c is false, so `t` is no thread-local register; (2) such temporaries
    live in scratch memory that is read where the condition does not hold: it must not hold stale bytes (a byte that is
    neither 0 nor 1 read as a C++ bool is undefined behaviour)."""
    import torch

    import oracle.numpy_backend  # noqa: F401
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from gt4py_amd.cartesian.gtscript import PARALLEL, Field, computation, interval  # noqa: F401

    def program(a: Field[np.float64], flag: Field[np.bool_], out: Field[np.float64]):  # noqa: F821
        with computation(PARALLEL), interval(...):
            t0 = a * 2.0
            t2 = a[1, 0, 0] + 1.0
            u = t0[1, 0, 0] - t0[-1, 0, 0]      # t0 is read at an offset below: a stage cut after its definition ...
            if flag and flag[0, 1, 0]:
                t0 = u
            elif flag and a > 0.0:              # (an `elif`: its mask temporary is assigned under the first condition only)
                t2 = u * 3.0                    # ... with `t2 = ...` above and this conditional update below the cut
            else:
                t2 = t2 - u
            out = t2 + t0[0, 1, 0]

    from gt4py_amd.cartesian.backend import stage_planner

    # without the substitution of conditionally assigned temporaries `t0` stays in memory and cuts the block in stages
    monkeypatch.setattr(stage_planner, "INLINE_MASKED", inline_masked)
    junk = torch.full((1 << 27,), 171, dtype=torch.uint8, device="cuda")  # whatever the scratch buffer gets is stale
    del junk
    ref = gtscript.stencil(backend="numpy", definition=program)
    hip = gtscript.stencil(backend="hip:mi300", definition=program, rebuild=True)
    plan = type(hip)._gt_program_.plan
    if inline_masked:
        assert len(plan.stages) == 1 and sorted(plan.scratch) == ["mask_1"], (len(plan.stages), sorted(plan.scratch))
    else:
        assert len(plan.stages) > 1 and "t2" in plan.scratch, (len(plan.stages), sorted(plan.scratch), sorted(plan.locals))
    domain = (70, 9, 4)
    arrays, origins = zoo.make_inputs(ref, domain, 5)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, origin=origins, domain=domain)
    dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
    hip(**dev, origin=origins, domain=domain)
    for k in arrays:
        np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=k)


@pytest.mark.parametrize("name", ["boundary_and_interior", "value_crosses_blocks", "if_else_pair", "if_else_with_interference",
                                  "condition_input_rewritten_between_branches"])
@pytest.mark.parametrize("domain", [(130, 11, 3), (5, 4, 2)])
def test_planner_case_programs_on_the_device(name, domain):
    """The programs written for the planner's rewriting passes (tests/planner_cases/programs.py; their plans are pinned in
    tests/test_codegen.py) run on the device: every field bit for bit against the oracle."""
    import oracle.numpy_backend  # noqa: F401
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from planner_cases import programs as P

    defn = getattr(P, name)
    ref = gtscript.stencil(backend="numpy", definition=defn)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn)
    arrays, origins = zoo.make_inputs(ref, domain, 17)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, origin=origins, domain=domain)
    dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
    hip(**dev, origin=origins, domain=domain)
    for k in arrays:
        np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"{name} {domain}: field {k}")


@pytest.mark.parametrize("domain", [(70, 5, 7), (66, 3, 58), (130, 4, 80), (64, 2, 161)])
def test_boundary_only_write_is_not_served_from_the_top_of_column_cache(domain):
    """`acc` is written at the first level only and read back on all levels by the second sweep; the
    `_tc` variants (K >= 58) used to read it from registers / LDS that the first sweep never filled."""
    import oracle.numpy_backend  # noqa: F401
    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian import gtscript
    from planner_cases import programs as P

    defn = P.boundary_only_write_read_back
    ref = gtscript.stencil(backend="numpy", definition=defn)
    hip = gtscript.stencil(backend="hip:mi300", definition=defn)
    arrays, origins = zoo.make_inputs(ref, domain, 23)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, origin=origins, domain=domain)
    dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
    hip(**dev, origin=origins, domain=domain)
    for k in arrays:
        np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"{domain}: field {k}")



# ---- vertical_advection_dycore against restatements that do not pass through this repository's frontend / IR -----------
VADV_FIELDS = ("utens_stage", "u_stage", "wcon", "u_pos", "utens")


def _vadv_stencil():
    import bench
    from gt4py_amd.cartesian import gtscript

    return gtscript.stencil(backend="hip:mi300", definition=bench._vertical_advection_dycore,
                            externals={"BET_M": 0.5, "BET_P": 0.5})


def test_vertical_advection_reproduces_the_golden_vector_of_the_reference_field_shim():
    """tests/golden/vadv_small.npz (scripts/make_golden.py: the numpy backend's statement code on the reference's own
    ``Field`` class, a different origin per field) through the generated kernels, bit for bit."""
    import pathlib

    import gt4py_amd.storage as gt_storage

    gold = np.load(pathlib.Path(__file__).parent / "golden" / "vadv_small.npz")
    origins = {n: tuple(int(v) for v in gold["vadv_origins"][k]) for k, n in enumerate(VADV_FIELDS)}
    dev = {n: gt_storage.from_array(gold["vadv_" + n], backend="hip:mi300", aligned_index=origins[n]) for n in VADV_FIELDS}
    _vadv_stencil()(**dev, dtr_stage=float(gold["vadv_dtr_stage"]), origin=origins, domain=tuple(int(v) for v in gold["vadv_domain"]))
    got = dev["utens_stage"].get()
    assert got.tobytes() == gold["vadv_utens_stage_out"].tobytes()


@pytest.mark.parametrize("domain", [(5, 4, 3), (70, 3, 57), (66, 5, 58), (64, 4, 80), (130, 3, 160), (64, 2, 161), (33, 2, 200)])
def test_vertical_advection_equals_the_independent_restatements(domain):
    """``oracle.ref_numpy.vadv`` (level-by-level slices) and, on the small domains, ``ref_debug_order.vadv_debug_order``
    (columns innermost) are written against the reference's definition and use neither the product's frontend nor its
    IR; the generated column kernels -- every rung of the top-of-column cache ladder these K depths select -- must give
    exactly their values."""
    import gt4py_amd.storage as gt_storage
    from oracle import ref_debug_order as DBG, ref_numpy as R

    rng = np.random.default_rng(sum(domain))
    shape = (domain[0] + 1, domain[1], domain[2] + 1)
    host = {n: rng.uniform(-1, 1, shape) for n in VADV_FIELDS}
    dev = {n: gt_storage.from_array(v, backend="hip:mi300") for n, v in host.items()}
    _vadv_stencil()(**dev, dtr_stage=3.0 / 20.0, origin=(0, 0, 0), domain=domain)
    want = {n: v.copy() for n, v in host.items()}
    R.vadv(*[want[n] for n in VADV_FIELDS], 3.0 / 20.0, domain=domain)
    for n in VADV_FIELDS:
        assert dev[n].get().tobytes() == want[n].tobytes(), n
    if domain[0] * domain[1] * domain[2] < 2000:
        twin = {n: v.copy() for n, v in host.items()}
        DBG.vadv_debug_order(*[twin[n] for n in VADV_FIELDS], 3.0 / 20.0, domain=domain)
        assert twin["utens_stage"].tobytes() == want["utens_stage"].tobytes()


def test_vertical_advection_at_the_bench_size():
    """1024 x 1024 x 160 (what bench.py times): columns from the corners and the middle of the domain bit for bit against
    ``ref_numpy.vadv`` run on exactly the data those columns read (a column reads its own levels and wcon of the column to
    its east), and on ALL columns the residual of the tridiagonal system the stencil assembles and solves, evaluated on the
    device with formulas that share nothing with the sweeps (cf. tests/test_oracle_restatements.py::vadv_residual)."""
    import torch

    import gt4py_amd.storage as gt_storage
    from oracle import ref_numpy as R

    domain = (1024, 1024, 160)
    di, dj, dk = domain
    shape = (di + 1, dj, dk + 1)
    gen = torch.Generator(device="cuda").manual_seed(77)
    dev = {}
    for n in VADV_FIELDS:
        f = gt_storage.empty(shape, np.float64, backend="hip:mi300", aligned_index=(0, 0, 0))
        f.tensor.copy_(torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1)
        dev[n] = f
    windows = [(0, 0), (1, 1023), (511, 300), (1022, 0), (1023, 1023), (700, 512)]
    before = {w: {n: dev[n].tensor[w[0]:w[0] + 2, w[1]:w[1] + 1, :].cpu().numpy() for n in VADV_FIELDS} for w in windows}
    dtr = 3.0  # diagonally dominant for wcon in [-1, 1): the residual bound below is meaningful
    old = {n: dev[n].tensor.clone() for n in ("utens_stage",)}
    _vadv_stencil()(**dev, dtr_stage=dtr, origin=(0, 0, 0), domain=domain)
    torch.cuda.synchronize()
    for w in windows:
        want = {n: v.copy() for n, v in before[w].items()}
        R.vadv(*[want[n] for n in VADV_FIELDS], dtr, domain=(1, 1, dk))
        got = dev["utens_stage"].tensor[w[0]:w[0] + 1, w[1]:w[1] + 1, :dk].cpu().numpy()
        assert got.tobytes() == want["utens_stage"][:1, :, :dk].tobytes(), w
    # residual on the device, all 1024 x 1024 columns
    w_ = dev["wcon"].tensor
    gav = -0.25 * (w_[1:, :, :dk] + w_[:-1, :, :dk])
    gcv = 0.25 * (w_[1:, :, 1:dk + 1] + w_[:-1, :, 1:dk + 1])
    a, c = gav * 0.5, gcv * 0.5
    del gav, gcv
    a[:, :, 0] = 0.0
    c[:, :, dk - 1] = 0.0
    b = dtr - a - c
    us = dev["u_stage"].tensor[:di]
    d = dtr * dev["u_pos"].tensor[:di, :, :dk] + dev["utens"].tensor[:di, :, :dk] + old["utens_stage"][:di, :, :dk]
    d[:, :, 1:] += -a[:, :, 1:] * (us[:, :, 0:dk - 1] - us[:, :, 1:dk])
    d[:, :, :dk - 1] += -c[:, :, :dk - 1] * (us[:, :, 1:dk] - us[:, :, 0:dk - 1])
    x = dev["utens_stage"].tensor[:di, :, :dk] / dtr + dev["u_pos"].tensor[:di, :, :dk]
    lhs = b * x
    lhs[:, :, 1:] += a[:, :, 1:] * x[:, :, :-1]
    lhs[:, :, :-1] += c[:, :, :-1] * x[:, :, 1:]
    rel = ((lhs - d).abs() / (d.abs() + (b * x).abs() + 1e-300)).max().item()
    assert rel < 1e-10, rel  # 160 levels of forward elimination and back substitution: a few 1e-12 (5 x 4 x 40: < 1e-13 on the CPU)
