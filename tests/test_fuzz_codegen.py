"""Differential fuzzing of the generic executor with random stencils (tests/fuzz_stencils.py).

CPU: the planner's rewritten IR (inlined temporaries, SSA versions) must evaluate, under the numpy oracle,
to the same bits as the original program, and the generated HIP must compile.  GPU: generated kernels vs
the oracle on every array.  Seeds are fixed, so a failure reproduces; the offending source is printed.
"""

import numpy as np
import pytest

import fuzz_stencils
import stencil_zoo as zoo

import os

# GT4MI_FUZZ_SEEDS=<n> widens both sweeps (one-off campaigns; results under profiles/)
_N = int(os.environ.get("GT4MI_FUZZ_SEEDS", "0"))
_FIRST = int(os.environ.get("GT4MI_FUZZ_FIRST_SEED", "0"))  # campaigns on fresh programs: seeds _FIRST .. _FIRST + _N - 1
CPU_SEEDS = list(range(_FIRST, _FIRST + (_N or 150)))
GPU_SEEDS = list(range(_FIRST, _FIRST + (_N or 200)))
DOMAINS = [(9, 7, 5), (66, 5, 4), (3, 3, 2)]


def _fit(domain, stencil_object, text):
    """Domains the program is valid on: K at least the stencil's minimum; with horizontal regions anchored
    at one edge (`: I[0] + 3`) the domain must be wider than the region's reach -- on a 3-wide domain such a
    region touches the far edge and its offset reads leave the array, in the reference just the same."""
    ni, nj, nk = domain
    if "region[" in text:
        ni, nj = max(ni, 6), max(nj, 6)
    return (ni, nj, max(nk, stencil_object.domain_info.min_sequential_axis_size))


def _build(seed, tmp_path, backend):
    import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
    from gt4py_amd.cartesian import gtscript

    defn, scalars, text = fuzz_stencils.make_stencil(seed, tmp_path)
    return gtscript.stencil(backend=backend, definition=defn), scalars, text


@pytest.mark.parametrize("seed", CPU_SEEDS)
def test_rewritten_ir_matches_original_under_the_oracle(seed, tmp_path):
    import oracle.numpy_backend as oracle_backend
    from gt4py_amd import _lib
    from gt4py_amd.cartesian import analysis

    ref, scalars, text = _build(seed, tmp_path, "numpy")
    hip, _, _ = _build(seed, tmp_path, "hip:mi300")
    program = getattr(type(hip), "_gt_program_", None)
    assert program is not None, text
    assert ref.field_info == hip.field_info
    domain = _fit(DOMAINS[seed % len(DOMAINS)], ref, text)
    arrays, origins = zoo.make_inputs(ref, domain, seed)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, **scalars, origin=origins, domain=domain)
    got = {k: v.copy() for k, v in arrays.items()}
    rewritten = program.plan.stencil
    oracle_backend.run_stencil(rewritten, analysis.compute_extents(rewritten), domain, origins, got, scalars)
    for k in arrays:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"seed {seed}, field {k}\n{text}")
    if seed % 6 == 0:  # compile a sample (hiprtc takes ~0.3 s per program)
        assert _lib.rtc_compile(program.source, f"fuzz_{seed}.hip", ["-DGT4MI_UNIT_I_STRIDE=1", "-DGT4MI_NO_ALIAS=1"])[:4] == b"\x7fELF"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", GPU_SEEDS)
def test_generated_kernels_match_the_oracle_on_random_stencils(seed, tmp_path):
    import gt4py_amd.storage as gt_storage

    ref, scalars, text = _build(seed, tmp_path, "numpy")
    hip, _, _ = _build(seed, tmp_path, "hip:mi300")
    for domain in (DOMAINS[seed % len(DOMAINS)], (130, 9, 6)):
        domain = _fit(domain, ref, text)
        arrays, origins = zoo.make_inputs(ref, domain, seed)
        expect = {k: v.copy() for k, v in arrays.items()}
        ref(**expect, **scalars, origin=origins, domain=domain)
        dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k],
                                        dimensions=tuple(hip.field_info[k].axes) + tuple(str(n) for n in range(len(hip.field_info[k].data_dims))))
               for k, v in arrays.items()}
        hip(**dev, **scalars, origin=origins, domain=domain)
        for k in arrays:
            np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"seed {seed} {domain}, field {k}\n{text}")


# ---- two-sweep column programs: the top-of-column cache of the code generator (stage_planner.TopCache) -----------
TWO_SWEEP_SEEDS = list(range(_FIRST, _FIRST + (_N or 60)))


def _two_sweep(seed, tmp_path, backend, **opts):
    import oracle.numpy_backend  # noqa: F401
    from gt4py_amd.cartesian import gtscript

    defn, scalars, text = fuzz_stencils.make_two_sweep_stencil(seed, tmp_path)
    return gtscript.stencil(backend=backend, definition=defn, **opts), scalars, text


@pytest.mark.parametrize("seed", TWO_SWEEP_SEEDS[::3])
def test_two_sweep_programs_plan_and_compile(seed, tmp_path):
    """CPU: most random Thomas-like programs qualify for the cache, and the `_tc` kernel compiles (shallow depths, so
    that every range is emitted)."""
    from gt4py_amd import _lib
    from gt4py_amd.cartesian.backend import hip_codegen

    saved = hip_codegen.TUNING["top_cache"]
    hip_codegen.TUNING["top_cache"] = (3, 4 * 8 * 3 * 256)
    try:
        hip, _, text = _two_sweep(seed, tmp_path, "hip:mi300", rebuild=True)
    finally:
        hip_codegen.TUNING["top_cache"] = saved
    program = type(hip)._gt_program_
    if program.plan.top_cache:
        assert any(k.top_cache is not None for k in program.kernels) and "_tc3(const gt_args a)" in program.source, text
    assert _lib.rtc_compile(program.source, f"two_sweep_{seed}.hip", ["-DGT4MI_UNIT_I_STRIDE=1", "-DGT4MI_NO_ALIAS=1"])[:4] == b"\x7fELF"


def test_most_two_sweep_programs_qualify(tmp_path):
    n = sum(bool(type(_two_sweep(seed, tmp_path, "hip:mi300")[0])._gt_program_.plan.top_cache) for seed in TWO_SWEEP_SEEDS[:30])
    assert n >= 20, f"only {n} of 30 random two-sweep programs are served by the top-of-column cache"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", TWO_SWEEP_SEEDS)
def test_two_sweep_programs_match_the_oracle(seed, tmp_path):
    """GPU: random Thomas-like programs with shallow, seed-dependent cache depths on domains around the smallest one the
    `_tc` variant accepts, and with the default depths on a deep domain; every field bit for bit."""
    import random

    import gt4py_amd.storage as gt_storage
    from gt4py_amd.cartesian.backend import hip_codegen

    rnd = random.Random(seed)
    ref, scalars, text = _two_sweep(seed, tmp_path, "numpy")
    for depths, levels in (((rnd.randint(0, 6), rnd.randint(0, 6)), None), (None, rnd.choice([64, 97, 130, 161]))):
        saved = hip_codegen.TUNING["top_cache"]
        if depths is not None:
            hip_codegen.TUNING["top_cache"] = (depths[0], depths[1] * 8 * 3 * 256, 64)
        try:
            hip, _, _ = _two_sweep(seed, tmp_path, "hip:mi300", rebuild=True)
        finally:
            hip_codegen.TUNING["top_cache"] = saved
        kern = type(hip)._gt_program_.kernels[0]
        k_values = [levels] if levels else sorted({max(ref.domain_info.min_sequential_axis_size, k) for k in
                                                   ((kern.top_cache[-1][2] if kern.top_cache else 8) + d for d in (-1, 0, 1, 7))})
        for nk in k_values:
            domain = (66, 3, nk)
            arrays, origins = zoo.make_inputs(ref, domain, seed)
            expect = {k: v.copy() for k, v in arrays.items()}
            ref(**expect, **scalars, origin=origins, domain=domain)
            dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
            hip(**dev, **scalars, origin=origins, domain=domain)
            for k in arrays:
                np.testing.assert_array_equal(dev[k].get(), expect[k],
                                              err_msg=f"seed {seed} depths {depths} {domain} (cache {kern.top_cache}), field {k}\n{text}")


# ---- chains of horizontally offset temporaries: the strip kernel that shares them between lanes (`_vecs`) -----------
SHARED_SEEDS = list(range(_FIRST, _FIRST + (_N or 60)))


def _shared(seed, tmp_path, backend, **opts):
    import oracle.numpy_backend  # noqa: F401 - registers backend "numpy"
    from gt4py_amd.cartesian import gtscript

    defn, scalars, text = fuzz_stencils.make_shared_temporaries_stencil(seed, tmp_path)
    return gtscript.stencil(backend=backend, definition=defn, **opts), scalars, text


@pytest.mark.parametrize("seed", SHARED_SEEDS[::3])
def test_shared_temporaries_programs_compile(seed, tmp_path):
    """CPU: most random chains of offset temporaries get a `_vecs` kernel, and it compiles for gfx950 (the fuzzer found a
    compiler bug here: a float shifted between lanes and widened next made the DPP-combine pass emit v_cvt_f64_f32_dpp,
    which the target cannot encode -- hence the opaque shift)."""
    from gt4py_amd import _lib

    hip, _, text = _shared(seed, tmp_path, "hip:mi300")
    program = type(hip)._gt_program_
    if any(k.shared_halo for k in program.kernels):
        assert "_vecs(const gt_args a)" in program.source and "gt_shift<" in program.source, text
    assert _lib.rtc_compile(program.source, f"shared_{seed}.hip", ["-DGT4MI_UNIT_I_STRIDE=1", "-DGT4MI_NO_ALIAS=1"])[:4] == b"\x7fELF"


def test_most_shared_temporaries_programs_qualify(tmp_path):
    n = sum(any(k.shared_halo for k in type(_shared(seed, tmp_path, "hip:mi300")[0])._gt_program_.kernels) for seed in SHARED_SEEDS[:30])
    assert n >= 15, f"only {n} of 30 random programs with offset temporaries get the strip kernel that shares them"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SHARED_SEEDS)
def test_shared_temporaries_programs_match_the_oracle(seed, tmp_path):
    """GPU: the same programs on domains of one, two and three waves in I with whole and partial 5-row strips; every
    field bit for bit against the oracle."""
    import gt4py_amd.storage as gt_storage

    ref, scalars, text = _shared(seed, tmp_path, "numpy")
    hip, _, _ = _shared(seed, tmp_path, "hip:mi300")
    for domain in ((130, 11, 2), (64, 5, 3), (300 + seed % 7, 8, 1)):
        domain = domain[:2] + (max(domain[2], ref.domain_info.min_sequential_axis_size),)
        arrays, origins = zoo.make_inputs(ref, domain, seed)
        expect = {k: v.copy() for k, v in arrays.items()}
        ref(**expect, **scalars, origin=origins, domain=domain)
        dev = {k: gt_storage.from_array(v, dtype=v.dtype, backend="hip:mi300", aligned_index=origins[k]) for k, v in arrays.items()}
        hip(**dev, **scalars, origin=origins, domain=domain)
        for k in arrays:
            np.testing.assert_array_equal(dev[k].get(), expect[k], err_msg=f"seed {seed} {domain}, field {k}\n{text}")


@pytest.mark.parametrize("seed", SHARED_SEEDS)
def test_shared_temporaries_rewritten_ir_matches_original_under_the_oracle(seed, tmp_path):
    """CPU: what the planner makes of these programs -- one temporary per interval block, conditionally assigned
    temporaries substituted as selects, everything inlined into one stage -- evaluated by the oracle against the
    original program, every field bit for bit."""
    import oracle.numpy_backend as oracle_backend
    from gt4py_amd.cartesian import analysis

    ref, scalars, text = _shared(seed, tmp_path, "numpy")
    hip, _, _ = _shared(seed, tmp_path, "hip:mi300")
    program = type(hip)._gt_program_
    domain = (9, 7, max(3, ref.domain_info.min_sequential_axis_size))
    arrays, origins = zoo.make_inputs(ref, domain, seed)
    expect = {k: v.copy() for k, v in arrays.items()}
    ref(**expect, **scalars, origin=origins, domain=domain)
    got = {k: v.copy() for k, v in arrays.items()}
    rewritten = program.plan.stencil
    oracle_backend.run_stencil(rewritten, analysis.compute_extents(rewritten), domain, origins, got, scalars)
    for k in arrays:
        np.testing.assert_array_equal(got[k], expect[k], err_msg=f"seed {seed}, field {k}\n{text}")
