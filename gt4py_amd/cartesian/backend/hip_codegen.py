"""Generic executor, host half: typed stencil IR -> stages -> HIP source for gfx950.

What the reference does for stencils on ``gt:gpu`` is a compiler tower (GTIR -> OIR -> optimisation
passes -> GridTools C++ -> nvcc; /root/reference/src/gt4py/cartesian/gtc/gtcpp/gtcpp_codegen.py,
backend/gtcpp_backend.py:109-166).  This module is a much smaller thing with the same contract --
*the values the numpy backend produces* (gtc/numpy/npir_codegen.py:205-318) -- for the sub-language the
frontend accepts (assignments of expression trees under ``computation``/``interval``):

(passes 1-3 live in ``stage_planner.py``, pass 4 -- emission -- here)

1. ``inline_horizontal_temporaries``: inside one PARALLEL interval block, temporaries that are read at
   horizontal offsets and depend only on never-written inputs are substituted into their readers
   (SSA over re-assigned names), i.e. recomputed at the shifted point.  The reference's counterpart is
   on-the-fly recomputation in ``OnTheFlyMerging``
   (gtc/passes/oir_optimizations/horizontal_execution_merging.py:135-330).  Values are unchanged: the
   same expression is evaluated on the same inputs, in the same dtype, without contraction.
2. ``plan_stages``: program order is cut into *stages* (one kernel launch each) wherever a value
   written by one column is read by another (read-after-write or write-after-read at a horizontal
   offset); inside a stage every (i, j) column is independent.  A stage is mapped ``ijk`` (one
   thread per point) when it only holds PARALLEL work without vertical dependencies on its own
   writes, else ``column`` (one thread per column, K loops inside -- the numpy backend's
   ``for k_ in range(k, K)``, npir_codegen.py:243-248, turned inside out, which is exact because the
   columns are independent).
3. temporaries become thread-local values when a value never leaves the iteration that defined it,
   global scratch (I-contiguous, extent-padded) otherwise; in ``column`` stages a value only read one
   level behind the sweep is additionally forwarded in a register (cf. the reference's K caches,
   gtc/passes/oir_optimizations/caches.py:92-143).
4. ``emit``: one ``extern "C" __global__`` kernel per stage taking ONE struct by value.

Every IR node carries its dtype and every conversion is an explicit ``Cast`` (frontend.py), so the C
expression tree is a transliteration; hiprtc gets ``-ffp-contract=off`` and no fast-math flags.
``**`` and transcendental functions go to the device math library and are NOT bit-identical to numpy.
"""

from __future__ import annotations

import ctypes
import hashlib
from dataclasses import dataclass, field, replace as _replace
from typing import Any, Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from .. import analysis, ir

Extent2 = analysis.Extent2


def _env_tuple(name: str, default: Tuple[int, ...]) -> Tuple[int, ...]:
    import os

    text = os.environ.get(name)
    return tuple(int(x) for x in text.split(",")) if text else default


#: launch geometry / unrolling of the generated kernels, from the sweep in
#: profiles/r1_codegen_sweep.log (env overrides are for tuning experiments)
TUNING = {
    # threads along I, J; K levels per thread; consecutive J rows per thread (unrolled: the compiler
    # then shares the row loads and the recomputed temporaries between neighbouring rows)
    "block_ijk": _env_tuple("GT4MI_CODEGEN_BLOCK_IJK", (64, 4, 1, 1)),
    "block_column": _env_tuple("GT4MI_CODEGEN_BLOCK_COLUMN", (64, 4)),
    "unroll": _env_tuple("GT4MI_CODEGEN_UNROLL", (8,))[0],  # sequential K loops
    # XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (private L2s); with R > 0 each
    # XCD gets runs of R consecutive tile rows, so rows shared by neighbouring tiles hit in one L2
    # (measured neutral-to-negative for thread-per-point kernels, profiles/r1_codegen_sweep.log: off)
    "xcd_rows": _env_tuple("GT4MI_CODEGEN_XCD_ROWS", (0,))[0],
    "nontemporal": _env_tuple("GT4MI_CODEGEN_NONTEMPORAL", (1,))[0],  # streaming stores for write-only outputs
    # column stages: issue the loads of up to this many K levels ahead of the dependent arithmetic (0 = off),
    # as long as the chunk needs no more than `prefetch_loads` values in flight per thread
    "prefetch": _env_tuple("GT4MI_CODEGEN_PREFETCH", (8,))[0],
    "prefetch_loads": _env_tuple("GT4MI_CODEGEN_PREFETCH_LOADS", (40,))[0],
    "vector": _env_tuple("GT4MI_CODEGEN_VECTOR", (1,))[0],  # 16-byte lanes for horizontal stages
    # ... and consecutive J rows per lane in those kernels: rows (and recomputed temporaries) shared by
    # neighbouring output rows are loaded (computed) once per strip
    "vector_rows": _env_tuple("GT4MI_CODEGEN_VECTOR_ROWS", (4,))[0],
    # strip kernels of stages whose temporaries were inlined: compute every temporary ONCE per point and pass it to the
    # neighbouring lanes with DPP shifts (waves overlap by a halo lane or two) instead of re-deriving it at every
    # offset it is read at (1), or only the recomputing form (0)
    "shared_temporaries": _env_tuple("GT4MI_CODEGEN_SHARED_TEMPORARIES", (1,))[0],
    # J rows per lane of that kernel (0 = by element size: 4 rows for 8-byte elements, 8 for 4-byte ones) and its XCD-aware
    # tile order (see xcd_rows).  Measured on the horizontal diffusion with XCD runs of 4, fp64 / fp32 GLUPS: rows 4 / 5 / 6 /
    # 7 / 8 = 226 / 221 / 218 / 212 / 219 and 386 / 390 / 403 / 410 / 424; runs of 2 / 4 / 8 tile rows differ by < 1 %, no
    # grouping costs 2 % (profiles/r2_codegen_shared_rows_xcd.log, r2_codegen_shared_xcd.log)
    "shared_rows": _env_tuple("GT4MI_CODEGEN_SHARED_ROWS", (0,))[0],
    "shared_xcd_rows": _env_tuple("GT4MI_CODEGEN_SHARED_XCD_ROWS", (4,))[0],
    # two-sweep column stages (stage_planner.TopCache): levels of the forward sweep's results kept in registers and,
    # below those, in LDS for the backward sweep -- (register levels, LDS bytes per workgroup, cap on the LDS levels).
    # Register levels < 0 (the default): one `_tc<n>` kernel per depth n = n_max, n_max - 8, n_max - 8 - step, ... and 16,
    # where n_max is what `top_cache_auto` = (register budget in dwords per lane, step in levels) allows for the cached
    # fields of the stage (vertical advection: 2 fp64 fields = 4 dwords per level -> 112, 104, 80, 56, 32, 16 levels + 40
    # in LDS, i.e. smallest K 154, 146, 122, 98, 74, 58: K = 80 keeps 72 levels on chip, K = 137 120, K = 60 56; 112
    # fits for it, the generated tridiagonal solve spills there and runs the 104-level variant); the
    # host launches the deepest variant the domain's K has room for and that compiled without spilling
    # (hip_generic._Variant).  A lone wave per SIMD owns 512 registers; a cached level costs exactly its dwords once the
    # register levels are pinned (_pin_register_level) and the second sweep has its own bases (_second_sweep_bases) --
    # before that it cost three times as much and 24 levels already spilled (profiles/r2_codegen_top_cache_deep_*.log).
    # (0, 0) = off; an explicit depth (GT4MI_CODEGEN_TOP_CACHE=80,163840) emits that one variant only.
    "top_cache": _env_tuple("GT4MI_CODEGEN_TOP_CACHE", (-1, 160 * 1024, 64)),
    "top_cache_auto": _env_tuple("GT4MI_CODEGEN_TOP_CACHE_AUTO", (448, 24)),
    # register levels of a `_tc` kernel: 0 = batches, each batch's loads right before its own arithmetic; 1 = the loads
    # of batch n + 1 issued before the arithmetic of batch n (+1..3 %, but a second batch of registers: as much as 10
    # cached levels are worth, profiles/r2_codegen_top_cache_pipeline.log); 2 = rolling: level n issues the loads of
    # level n + D and computes itself, D levels of loads always in flight for the registers of ONE more level --
    # vertical advection +8 % at equal depth, +3..5 % with the ladder (its 112-level variant no longer fits, 104 runs),
    # the generated tridiagonal solve unchanged (profiles/r2_codegen_top_cache_rolling.log, _onchip.log).
    "top_cache_pipeline": _env_tuple("GT4MI_CODEGEN_TOP_CACHE_PIPELINE", (2,))[0],
    "top_cache_lookahead": _env_tuple("GT4MI_CODEGEN_TOP_CACHE_LOOKAHEAD", (0,))[0],  # rolling prefetch distance (0: chunk depth)
    # the LDS levels of a `_tc` kernel as straight-line code like the register levels (1) or as a loop (0): within the
    # noise at K = 160 / 80 / 60 for 5x the source (profiles/r2_codegen_top_cache_onchip.log)
    "top_cache_unroll_lds": _env_tuple("GT4MI_CODEGEN_TOP_CACHE_UNROLL_LDS", (0,))[0],
    # `_tc` kernels: streaming (nontemporal) stores for what the kernel itself never reads back from memory -- the
    # second sweep's results, and the store-through copies of cached in/out fields at the levels that stay on chip
    "top_cache_streaming": _env_tuple("GT4MI_CODEGEN_TOP_CACHE_STREAMING", (1,))[0],
    # column kernels: nontemporal LOADS for what a column kernel reads exactly once (level after level, planes apart: nothing worth
    # keeping in the L1).  0 = plain loads; 1 = every load from memory; 2 = only of fields the stage reads at no horizontal offset;
    # 3 = ... that, moreover, only ONE of the stage's sweeps reads from memory; 4 = ... and that the stage does not write;
    # 5 (default) = 3 + fields BOTH sweeps read and nobody writes: cacheable in the first sweep, nontemporal in the last (their last use).
    # Same process, same fields, 1024 x 1024 x 160 fp64 (profiles/r5_nt_loads_column_kernels.txt): vertical advection 0.588 (0) /
    # 0.557 (1) / 0.584 (2) / **0.631 (3)** / 0.586 (4) of the HBM peak -- `u_pos`, which both sweeps read, must stay cacheable --;
    # generated tridiagonal solve 0.686 / 0.730 / 0.730 / **0.730** / 0.698; hand-written solve (tridiag_stack.hip.h NTL) +5-9 %.
    # 5 against 3 on another box: vertical advection 0.6456 -> 0.6476 (0.605 plain), the solve unchanged.
    "column_nt_loads": _env_tuple("GT4MI_CODEGEN_COLUMN_NT_LOADS", (5,))[0],
    # ... and in the 16-byte-lane strip kernels of horizontal stages: the arrays the stage reads at its own point only (_read_once_fields)
    "strip_nt_loads": _env_tuple("GT4MI_CODEGEN_STRIP_NT_LOADS", (1,))[0],
}

from .stage_planner import (Nest, Plan, Stage, Stmt, UnsupportedStencil, _field_reads, _stmt_field_reads,  # noqa: F401
                            inline_horizontal_temporaries, plan_stages)
from .stage_planner import _target_exprs as stage_planner_target_exprs


# ---------------------------------------------------------------------------------------------------
# 4. emission
# ---------------------------------------------------------------------------------------------------
_CTYPE = {
    "bool": "bool", "int8": "signed char", "int16": "short", "int32": "int", "int64": "long long",
    "uint8": "unsigned char", "uint16": "unsigned short", "uint32": "unsigned int", "uint64": "unsigned long long",
    "float32": "float", "float64": "double",
}
_CTYPES_TYPE = {
    "bool": ctypes.c_bool, "int8": ctypes.c_int8, "int16": ctypes.c_int16, "int32": ctypes.c_int32,
    "int64": ctypes.c_int64, "uint8": ctypes.c_uint8, "uint16": ctypes.c_uint16, "uint32": ctypes.c_uint32,
    "uint64": ctypes.c_uint64, "float32": ctypes.c_float, "float64": ctypes.c_double,
}

PRELUDE = r"""
// generated by gt4py_amd.cartesian.backend.hip_codegen -- do not edit
#pragma clang fp contract(off)
typedef long long gt_i64;
#define GT_DEV static __device__ __forceinline__
GT_DEV double gt_min(double a, double b) { return (a != a || b != b) ? a + b : (a < b ? a : b); }
GT_DEV float gt_min(float a, float b) { return (a != a || b != b) ? a + b : (a < b ? a : b); }
GT_DEV double gt_max(double a, double b) { return (a != a || b != b) ? a + b : (a > b ? a : b); }
GT_DEV float gt_max(float a, float b) { return (a != a || b != b) ? a + b : (a > b ? a : b); }
template <class T> GT_DEV T gt_min(T a, T b) { return a < b ? a : b; }
template <class T> GT_DEV T gt_max(T a, T b) { return a > b ? a : b; }
// numpy.remainder: result takes the sign of the divisor (npy_divmod)
GT_DEV double gt_mod(double a, double b) {
    double m = __builtin_fmod(a, b);
    if (b == 0.0) return m;
    if (m != 0.0) { if ((b < 0.0) != (m < 0.0)) m += b; } else { m = __builtin_copysign(0.0, b); }
    return m;
}
GT_DEV float gt_mod(float a, float b) {
    float m = __builtin_fmodf(a, b);
    if (b == 0.0f) return m;
    if (m != 0.0f) { if ((b < 0.0f) != (m < 0.0f)) m += b; } else { m = __builtin_copysignf(0.0f, b); }
    return m;
}
template <class T> GT_DEV T gt_mod(T a, T b) {
    if (b == 0) return 0;
    T m = a % b;
    return (m != 0 && ((m < 0) != (b < 0))) ? m + b : m;
}
// device-library entry points the compiler has no builtin lowering for on amdgcn
extern "C" __device__ double __ocml_pow_f64(double, double);
extern "C" __device__ float __ocml_pow_f32(float, float);
extern "C" __device__ double __ocml_erf_f64(double);
extern "C" __device__ float __ocml_erf_f32(float);
extern "C" __device__ double __ocml_erfc_f64(double);
extern "C" __device__ float __ocml_erfc_f32(float);
extern "C" __device__ double __ocml_tgamma_f64(double);
extern "C" __device__ float __ocml_tgamma_f32(float);
GT_DEV double gt_pow(double a, double b) { return __ocml_pow_f64(a, b); }
GT_DEV float gt_pow(float a, float b) { return __ocml_pow_f32(a, b); }
template <class T> GT_DEV T gt_pow(T a, T b) {
    T r = 1;
    for (T n = 0; n < b; ++n) r *= a;
    return r;
}
GT_DEV double gt_abs(double a) { return __builtin_fabs(a); }
GT_DEV float gt_abs(float a) { return __builtin_fabsf(a); }
template <class T> GT_DEV T gt_abs(T a) { return a < 0 ? -a : a; }
#define GT_MATH1(name, fd, ff) \
    GT_DEV double gt_##name(double a) { return fd(a); } \
    GT_DEV float gt_##name(float a) { return ff(a); }
// transcendental functions: the device library's (the ones HIP's <cmath> calls); the compiler's own builtins
// either have no lowering on amdgcn (tan, pow, erf ...) or lower to the low-accuracy hardware instructions
#define GT_OCML1(name) \
    extern "C" __device__ double __ocml_##name##_f64(double); \
    extern "C" __device__ float __ocml_##name##_f32(float); \
    GT_DEV double gt_##name(double a) { return __ocml_##name##_f64(a); } \
    GT_DEV float gt_##name(float a) { return __ocml_##name##_f32(a); }
GT_MATH1(sqrt, __builtin_sqrt, __builtin_sqrtf)
GT_MATH1(floor, __builtin_floor, __builtin_floorf)
GT_MATH1(ceil, __builtin_ceil, __builtin_ceilf)
GT_MATH1(trunc, __builtin_trunc, __builtin_truncf)
GT_OCML1(sin)
GT_OCML1(cos)
GT_OCML1(tan)
GT_OCML1(asin)
GT_OCML1(acos)
GT_OCML1(atan)
GT_OCML1(sinh)
GT_OCML1(cosh)
GT_OCML1(tanh)
GT_OCML1(asinh)
GT_OCML1(acosh)
GT_OCML1(atanh)
GT_OCML1(exp)
GT_OCML1(log)
GT_OCML1(log10)
GT_OCML1(cbrt)
GT_MATH1(erf, __ocml_erf_f64, __ocml_erf_f32)
GT_MATH1(erfc, __ocml_erfc_f64, __ocml_erfc_f32)
GT_MATH1(gamma, __ocml_tgamma_f64, __ocml_tgamma_f32)
GT_MATH1(round, __builtin_rint, __builtin_rintf)  // ties to even, like np.round / std::nearbyint
// ties away from zero the way the reference's numpy backend spells it (gtc/ufuncs.py:31-33)
GT_DEV double gt_round_away_from_zero(double a) { return __builtin_copysign(__builtin_floor(__builtin_fabs(a) + 0.5), a); }
GT_DEV float gt_round_away_from_zero(float a) { return __builtin_copysignf(__builtin_floorf(__builtin_fabsf(a) + 0.5f), a); }
// Workgroup -> tile.  The dispatcher deals workgroups round-robin to the 8 XCDs in linear order (x
// fastest); give each XCD runs of `rows` consecutive tile rows instead of every 8th tile.
GT_DEV void gt_tile(unsigned rows, unsigned& bx, unsigned& by, unsigned& bz) {
    if (rows == 0u) {  // no grouping: the hardware's own order
        bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
        return;
    }
    const unsigned long long gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    if (n < 0x80000000ull) {
        // 32-bit index arithmetic (every launch in practice): a 64-bit division is ~100 scalar instructions, and a strip
        // kernel's workgroup lives for a few microseconds -- five of them in front of the first load cost the generated
        // Laplacian 2-3 % (profiles/r3_codegen_strip_blocks.log)
        const unsigned ux = gridDim.x, uy = gridDim.y, un = (unsigned)n;
        unsigned l = blockIdx.x + ux * (blockIdx.y + uy * blockIdx.z);
        const unsigned g = ux * rows, span = 8u * g;
        if (l < (un / span) * span) {
            const unsigned xcd = l & 7u, slot = l >> 3, run = slot / g;
            l = (run * 8u + xcd) * g + (slot - run * g);
        }
        const unsigned row = l / ux;
        bx = l - row * ux;
        bz = row / uy;
        by = row - bz * uy;
        return;
    }
    unsigned long long l = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned long long g = gx * rows, span = 8ull * g;
    if (l < (n / span) * span) {
        const unsigned long long xcd = l % 8ull, slot = l / 8ull;
        l = ((slot / g) * 8ull + xcd) * g + slot % g;
    }
    bx = (unsigned)(l % gx);
    by = (unsigned)((l / gx) % gy);
    bz = (unsigned)(l / (gx * gy));
}
// 16-byte lanes: a lane owns N consecutive I points; the points next to them sit in the neighbouring
// lanes' registers and are fetched with whole-wave DPP shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1,
// no LDS).  bound_ctrl: lanes shifted in from outside the wave read 0 without an extra move -- callers overwrite those lanes.
template <class T, int N> using gt_vec = T __attribute__((ext_vector_type(N)));
// OPAQUE: the result goes through an empty asm, so that the DPP-combine pass cannot fold the move into its consumer.
// It does that to a float that is widened next (v_cvt_f64_f32_dpp), which gfx950 cannot encode with a wave shift ("DP ALU
// dpp only support row_newbcast": the compilation fails; found by the fuzzer).  The `_vecs` kernels use it; the `_vec`
// kernels overwrite the edge lanes' value after the shift, which keeps the pass away as a side effect.
template <class T, bool FROM_BELOW, bool OPAQUE = false> GT_DEV T gt_shift(T v) {
    if constexpr (sizeof(T) == 4) {
        const int b = __builtin_bit_cast(int, v);
        int r = FROM_BELOW ? __builtin_amdgcn_update_dpp(0, b, 0x138, 0xF, 0xF, true)
                           : __builtin_amdgcn_update_dpp(0, b, 0x130, 0xF, 0xF, true);
        if constexpr (OPAQUE) asm("" : "+v"(r));
        return __builtin_bit_cast(T, r);
    } else {
        static_assert(sizeof(T) == 8, "gt_shift: 4- or 8-byte types");
        const long long b = __builtin_bit_cast(long long, v);
        int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
        lo = FROM_BELOW ? __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true)
                        : __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xF, 0xF, true);
        hi = FROM_BELOW ? __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true)
                        : __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xF, 0xF, true);
        if constexpr (OPAQUE) asm("" : "+v"(lo), "+v"(hi));
        return __builtin_bit_cast(T, ((long long)hi << 32) | (unsigned int)lo);
    }
}
template <class T> GT_DEV bool gt_isnan(T a) { return a != a; }
template <class T> GT_DEV bool gt_isinf(T a) { return a == a && (a - a) != (a - a); }
template <class T> GT_DEV bool gt_isfinite(T a) { return (a - a) == (a - a); }
"""

_EXACT_CALLS = {"abs", "min", "max", "mod", "sqrt", "floor", "ceil", "trunc", "isfinite", "isinf", "isnan", "round",
                "round_away_from_zero"}


def _c_ident(name: str) -> str:
    return "".join(ch if ch.isalnum() else "_" for ch in name)


def _literal(value, dtype: np.dtype) -> str:
    dt = np.dtype(dtype)
    if dt == np.dtype("bool"):
        return "true" if value else "false"
    if dt.kind == "f":
        v = dt.type(value)
        if np.isnan(v):
            return f"(({_CTYPE[dt.name]})__builtin_nan(\"\"))"
        if np.isinf(v):
            return f"(({_CTYPE[dt.name]})({'-' if v < 0 else ''}__builtin_inf()))"
        text = float(v).hex()  # exact for float32 too: every float32 is a double
        return f"{text}f" if dt == np.dtype("float32") else text
    suffix = {"int64": "LL", "uint64": "ULL", "uint32": "U"}.get(dt.name, "")
    return f"(({_CTYPE[dt.name]}){int(value)}{suffix})"


@dataclass
class KernelSource:
    name: str
    mapping: str
    extent: Extent2
    block: Tuple[int, int, int]
    k_per_thread: int = 1
    j_per_thread: int = 1
    vec: int = 0  # > 0: a `<name>_vec` kernel exists in which every lane owns `vec` consecutive I points
    vec_fields: Tuple[str, ...] = ()  # arrays whose alignment / strides decide whether it may be launched
    vec_rows: int = 1  # consecutive J rows per lane in the `_vec` kernel
    plane: Optional[Tuple[int, str, ir.Interval]] = None  # launched once per K level by the host (Stage.plane)
    #: ((register levels, LDS levels, smallest domain K), ...), deepest first: for each a `<name>_tc<register levels>`
    #: kernel exists that keeps that many top levels of a two-sweep column stage on chip; the host launches the first
    #: one the domain has enough levels for
    top_cache: Optional[Tuple[Tuple[int, int, int], ...]] = None
    #: halo lanes per side of the `<name>_vecs` kernel (strip kernel whose temporaries are computed once per point and
    #: passed between lanes), or 0: none.  Its waves overlap by 2 * shared_halo lanes; the host sizes the grid for
    #: (64 - 2 * shared_halo) * vec output columns per wave.
    shared_halo: int = 0
    shared_rows: int = 0  # J rows per lane of the `_vecs` kernel
    shared_vec: int = 0  # I points per lane of the `_vecs` kernel (it may exist where no `_vec` kernel does)
    shared_fields: Tuple[str, ...] = ()  # arrays whose alignment / strides decide whether it may be launched
    #: launch `_vecs` rather than `_vec` when both exist: yes when the shared temporaries are 8-byte values (their
    #: arithmetic is what binds the recomputing kernel); with 4-byte temporaries recomputing is cheaper than passing
    #: them around (all-float32 hdiff: 466 GLUPS recomputing, 441 sharing; fp64 internals: 418 and 434)
    shared_preferred: bool = False
    #: workgroup shape of the `_vec` kernel when it differs from `block` (light stages: 256 x 1, see _strip_shape)
    vec_block: Optional[Tuple[int, int, int]] = None


@dataclass
class GeneratedProgram:
    source: str
    kernels: List[KernelSource]
    args_struct: type  # ctypes.Structure subclass mirroring `struct gt_args`
    plan: Plan
    inexact_calls: Set[str]

    @property
    def digest(self) -> str:
        return hashlib.sha256(self.source.encode()).hexdigest()[:16]


class _Emitter:
    def __init__(self, plan: Plan):
        self.plan = plan
        self.lines: List[str] = []
        self.inexact: Set[str] = set()
        self.decl_dtype: Dict[str, np.dtype] = {}
        for d in (*plan.stencil.fields, *plan.stencil.temporaries):
            self.decl_dtype[d.name] = np.dtype(d.dtype)
        self.axes: Dict[str, Tuple[str, ...]] = {f.name: tuple(f.axes) for f in (*plan.stencil.fields, *plan.stencil.temporaries)}
        self.data_dims: Dict[str, Tuple[int, ...]] = {d.name: tuple(d.data_dims) for d in
                                                      (*plan.stencil.fields, *plan.stencil.temporaries)}
        self.global_names = [f.name for f in plan.api_fields] + list(plan.scratch)
        self.written = {s.target.name for _, _, s in plan.stencil.statements()}
        read = {e.name for _, _, s in plan.stencil.statements() for e in ir.stmt_reads(s) if isinstance(e, ir.FieldAccess)}
        api = {f.name for f in plan.api_fields}
        # outputs nobody reads back: keep them out of the caches
        self.streaming = (self.written & api) - read if TUNING["nontemporal"] else set()
        # state of the 16-byte-lane emission (vector_kernel)
        self.vec_rows: Optional[Dict[Tuple[str, int, int], Dict[int, str]]] = None
        self.vec_component = 0
        self.vec_row = 0
        self.local_suffix = ""
        self.base_prefix = "b_"
        self.shared_lookup = None  # strip kernel with shared temporaries: FieldAccess -> register, or None
        # loads issued ahead of a chunk of K levels: (name, offset, data index) -> register, per statement
        self.prefetched: Dict[Tuple, str] = {}
        self.prefetch_for: Dict[int, Dict[Tuple, str]] = {}
        # top-of-column cache (kernel variant `_tc`): None, or (TopCache, register levels, LDS levels, threads per
        # workgroup), and where the level being emitted lives: ("mem",) | ("lds",) | ("reg", slot)
        self.tc: Optional[Tuple] = None
        self.tc_mode: Tuple = ("mem",)
        self.tc_sweep2 = False  # emitting the backward sweep: every cached field is complete, reads come from the cache
        self.tc_written: Set[str] = set()  # forward sweep: cached fields already assigned at the level being emitted
        self.nt_loads: Optional[Set[str]] = None  # column kernels: the fields whose loads from memory are nontemporal
        self.nt_loads_last: Dict[str, Any] = {}  # ... and fields whose loads are nontemporal only in the sweep of this order (their last use)
        self.nt_nest_order = None  # the loop order of the nest being emitted

    # -- expressions --------------------------------------------------------------------------
    def access(self, e: ir.FieldAccess, k: str, stage_index: int, reg: Dict[str, str], store: bool = False) -> str:
        """C expression of a field access.  ``reg`` maps (name, k offset) -> register holding that level; with
        ``store`` the expression is the memory location itself (an lvalue), whatever is held in registers."""
        name = e.name
        if not store and self.shared_lookup is not None:
            hit = self.shared_lookup(e)
            if hit is not None:
                return hit
        if name in self.plan.locals:
            return f"l_{_c_ident(name)}{self.local_suffix}"
        if not store:
            if self.prefetched and e.koffset is None:
                hit = self.prefetched.get((name, tuple(e.offset), tuple(e.data_index or ())))
                if hit is not None:
                    return hit
            if self.vec_rows is not None and "I" in self.axes.get(name, ("I", "J", "K")):
                return self.vec_rows[(name, e.offset[1] + self.vec_row, e.offset[2])][self.vec_component + e.offset[0]]
            if (name, e.offset[2]) in reg and e.offset[:2] == (0, 0) and e.koffset is None:
                return reg[(name, e.offset[2])]
        if name in self.plan.register_only:
            raise AssertionError(f"register-only temporary '{name}' needs memory at offset {e.offset}")
        c = _c_ident(name)
        if (not store and self.tc is not None and self.tc_mode[0] != "mem" and name in self.tc[0].names
                and tuple(e.offset) == (0, 0, 0) and e.koffset is None and (self.tc_sweep2 or name in self.tc_written)):
            # the level lives on chip.  (Forward sweep: only once this level has been assigned -- an in/out field is
            # first READ at the level, and that old value is in memory.)
            return self.tc_slot(name, k)
        di, dj, dk = e.offset
        axes = self.axes.get(name, ("I", "J", "K"))
        terms = []
        if "K" in axes:
            if e.koffset is not None:
                saved_rows, self.vec_rows = self.vec_rows, None  # the index itself is read from memory
                level = f"(gt_i64)({self.expr(e.koffset, k, stage_index, reg)})"
                # absolute: counted from the field's K origin, which is where the base pointer stands
                terms.append(f"({level}) * a.{c}_sk" if e.absolute_k else f"({k}{dk:+d} + {level}) * a.{c}_sk")
                self.vec_rows = saved_rows
            else:
                terms.append(f"({k}{dk:+d}) * a.{c}_sk" if dk else f"{k} * a.{c}_sk")
        if di and "I" in axes:
            terms.append(f"{di} * GT_SI(a.{c}_si)")
        if dj and "J" in axes:
            terms.append(f"{dj} * a.{c}_sj")
        for n, d in enumerate(e.data_index or ()):
            if isinstance(d, ir.Expr):  # run-time data index
                terms.append(f"(gt_i64)({self.expr(d, k, stage_index, reg)}) * a.{c}_d{n}")
            elif d:
                terms.append(f"{d} * a.{c}_d{n}")
        ref = f"{self.base_prefix}{c}[{' + '.join(terms) if terms else '0'}]"
        if not store and self.nt_loads is not None and (name in self.nt_loads or (name in self.nt_loads_last
                                                                                   and self.nt_loads_last[name] is self.nt_nest_order)):
            return f"__builtin_nontemporal_load(&{ref})"
        return ref

    def expr(self, e: ir.Expr, k: str, si: int, reg: Dict[str, str]) -> str:
        rec = lambda x: self.expr(x, k, si, reg)  # noqa: E731
        if isinstance(e, ir.Literal):
            return _literal(e.value, e.dtype)
        if isinstance(e, ir.FieldAccess):
            return self.access(e, k, si, reg)
        if isinstance(e, ir.ScalarAccess):
            return f"a.p_{_c_ident(e.name)}"
        if isinstance(e, ir.AxisIndex):  # the K level, counted from the start of the compute domain
            return f"(({_CTYPE[np.dtype(e.dtype).name]})({k}))"
        if isinstance(e, ir.Cast):
            dt = np.dtype(e.dtype)
            inner = rec(e.expr)
            return f"(({inner}) != 0)" if dt == np.dtype("bool") else f"(({_CTYPE[dt.name]})({inner}))"
        ct = _CTYPE[np.dtype(e.dtype).name] if getattr(e, "dtype", None) is not None else None
        if isinstance(e, ir.UnaryOp):
            if e.op == "not":
                return f"(!({rec(e.expr)}))"
            return f"(({ct})({e.op}({rec(e.expr)})))"
        if isinstance(e, ir.BinaryOp):
            left, right = rec(e.left), rec(e.right)
            if e.op in ("+", "-", "*", "/"):
                return f"(({ct})(({left}) {e.op} ({right})))"
            if e.op == "%":
                return f"gt_mod({left}, {right})"
            if e.op == "**":
                self.inexact.add("**")
                return f"gt_pow({left}, {right})"
            if e.op in ir.COMPARISON_OPS:
                return f"(({left}) {e.op} ({right}))"
            if e.op == "and":
                return f"((bool)({left}) & (bool)({right}))"
            if e.op == "or":
                return f"((bool)({left}) | (bool)({right}))"
            raise UnsupportedStencil(f"binary operator {e.op}")
        if isinstance(e, ir.TernaryOp):
            return f"(({rec(e.cond)}) ? ({rec(e.true_expr)}) : ({rec(e.false_expr)}))"
        if isinstance(e, ir.NativeCall):
            if e.func.startswith("cast:"):
                dt = np.dtype(e.dtype)
                inner = rec(e.args[0])
                return f"(({inner}) != 0)" if dt == np.dtype("bool") else f"(({_CTYPE[dt.name]})({inner}))"
            if e.func not in _EXACT_CALLS:
                self.inexact.add(e.func)
            return f"gt_{e.func}({', '.join(rec(a) for a in e.args)})"
        raise UnsupportedStencil(f"expression node {type(e).__name__}")

    # -- statements ---------------------------------------------------------------------------
    def guard(self, s: Stmt, stage: Stage, ivar: str = "i", jvar: str = "j") -> Optional[str]:
        (ilo, ihi), (jlo, jhi) = s.extent
        (silo, sihi), (sjlo, sjhi) = stage.extent
        conds = []
        if ilo != silo:
            conds.append(f"{ivar} >= {ilo}")
        if ihi != sihi:
            conds.append(f"{ivar} < a.dI + ({ihi})")
        if jlo != sjlo:
            conds.append(f"{jvar} >= {jlo}")
        if jhi != sjhi:
            conds.append(f"{jvar} < a.dJ + ({jhi})")
        return " && ".join(conds) if conds else None

    def full_guard(self, s: Stmt, stage: Stage, si: int, k: str, reg: Dict, ivar: str = "i", jvar: str = "j") -> Optional[str]:
        """Extent guard, horizontal-region mask and run-time mask of a statement as one condition."""
        g = self.guard(s, stage, ivar, jvar)
        if s.region is not None:  # horizontal mask: bounds relative to the compute domain
            conds = []
            for var, size, iv in ((ivar, "a.dI", s.region.i), (jvar, "a.dJ", s.region.j)):
                for b, op in ((iv.start, ">="), (iv.end, "<")):
                    if b is not None:
                        conds.append(f"{var} {op} {b.offset}" if b.level is ir.Level.START else f"{var} {op} {size} + ({b.offset})")
            if conds:
                r = " && ".join(conds)
                g = f"({g}) && {r}" if g else r
        if s.mask is not None:  # np.where(mask, value, target): untouched where the mask is false
            m = self.expr(s.mask, k, si, reg)
            g = f"({g}) && ({m})" if g else m
        return g

    def statement(self, s: Stmt, stage: Stage, si: int, k: str, reg: Dict, indent: str, carry: Sequence[str] = ()) -> None:
        self.prefetched = self.prefetch_for.get(id(s), {})
        try:
            self._statement(s, stage, si, k, reg, indent, carry)
        finally:
            self.prefetched = {}

    def _statement(self, s: Stmt, stage: Stage, si: int, k: str, reg: Dict, indent: str, carry: Sequence[str] = ()) -> None:
        value = self.expr(s.value, k, si, reg)
        name = s.target.name
        g = self.full_guard(s, stage, si, k, reg)
        pad = indent
        if g:
            self.lines.append(f"{indent}if ({g}) {{")
            pad = indent + "    "
        on_chip = self.tc is not None and self.tc_mode[0] != "mem" and name in self.tc[0].names
        if name in self.plan.locals:
            self.lines.append(f"{pad}l_{_c_ident(name)}{self.local_suffix} = {value};")
        elif name in carry:  # keep the freshly written level for this iteration's later reads and the next one
            self.lines.append(f"{pad}n_{_c_ident(name)} = {value};")
            if on_chip:
                self.lines.append(f"{pad}{self.tc_slot(name, k)} = n_{_c_ident(name)};")
            if name not in self.plan.register_only and (not on_chip or name in self.tc[0].store_through):
                self._global_store(s, k, si, reg, f"n_{_c_ident(name)}", pad, self._tc_streams(name, stage, on_chip))
            reg[(name, 0)] = f"n_{_c_ident(name)}"
        elif on_chip:
            self.lines.append(f"{pad}{self.tc_slot(name, k)} = {value};")
            if name in self.tc[0].store_through:
                self._global_store(s, k, si, reg, self.tc_slot(name, k), pad, self._tc_streams(name, stage, True))
        elif name in self.streaming or self._tc_streams(name, stage, False):
            self._global_store(s, k, si, reg, value, pad, True)
        else:
            self.lines.append(f"{pad}{self.access(s.target, k, si, reg, store=True)} = {value};")
        if on_chip:
            self.tc_written.add(name)
        if g:
            self.lines.append(f"{indent}}}")

    def _global_store(self, s: Stmt, k: str, si: int, reg: Dict, value: str, pad: str, streaming: bool) -> None:
        target = self.access(s.target, k, si, reg, store=True)
        if streaming:
            self.lines.append(f"{pad}__builtin_nontemporal_store({value}, &{target});")
        else:
            self.lines.append(f"{pad}{target} = {value};")

    def _tc_streams(self, name: str, stage: Stage, on_chip: bool) -> bool:
        """In a `_tc` kernel: may this store bypass the caches?  Yes for a level the kernel never loads again: any store
        of the second sweep to a field that sweep does not read at another level, and the first sweep's store-through
        copy of a cached field at a level that stays on chip (the second sweep reads the slot, not memory)."""
        if self.tc is None or not TUNING["top_cache_streaming"] or name in self.plan.scratch:
            return False
        first = self.tc[0].first_sweep_nests
        second_reads_elsewhere = any(
            e.name == name and (tuple(e.offset) != (0, 0, 0) or e.koffset is not None)
            for nest in stage.nests[first:] for st in nest.stmts for e in _stmt_field_reads(st))
        if self.tc_sweep2:
            return not second_reads_elsewhere
        return on_chip and not second_reads_elsewhere

    def statements(self, stmts: Sequence[Stmt], stage: Stage, si: int, k: str, reg: Dict, indent: str,
                   carry: Sequence[str] = ()) -> None:
        """Emit a statement list, opening and closing the `while` loops they sit in.  Per point a loop is
        `while (cond) { body }`: cond already carries the enclosing masks (ir.Assign.loops), and it is only
        evaluated where the loop's statements may run at all (their extent / region guard)."""
        open_loops: List[int] = []
        for s in stmts:
            ids = [lid for lid, _ in s.loops]
            common = 0
            while common < len(open_loops) and common < len(ids) and open_loops[common] == ids[common]:
                common += 1
            while len(open_loops) > common:
                open_loops.pop()
                self.lines.append(f"{indent}{'    ' * len(open_loops)}}}")
            for lid, cond in s.loops[common:]:
                pad = indent + "    " * len(open_loops)
                g = self.guard(s, stage)
                if s.region is not None:
                    only_region = Stmt(s.target, s.value, s.extent, None, s.region)
                    g = self.full_guard(only_region, stage, si, k, reg)
                c = self.expr(cond, k, si, reg)
                self.lines.append(f"{pad}while ({'(' + g + ') && ' if g else ''}({c})) {{")
                open_loops.append(lid)
            self.statement(s, stage, si, k, reg, indent + "    " * len(open_loops), carry)
        while open_loops:
            open_loops.pop()
            self.lines.append(f"{indent}{'    ' * len(open_loops)}}}")

    def tc_slot(self, name: str, k: str) -> str:
        """Where level ``k`` of a cached field lives in the current range of a `_tc` kernel (an lvalue)."""
        cache, n_reg, n_lds, threads = self.tc
        c = _c_ident(name)
        if self.tc_mode[0] == "reg":
            slot = self.tc_mode[1]
            if slot < 0:  # an LDS level of the straight-line on-chip range (top_cache_unroll_lds): constant index
                return f"tc_lds_{c}[{(slot + n_lds) * threads} + tc_tid]"
            return f"tc_{c}_{slot}"
        return f"tc_lds_{c}[(({k}) - (a.dK - {n_reg + n_lds})) * {threads} + tc_tid]"

    def column_in_extent(self, name: str, stage: Stage) -> Optional[str]:
        """Condition for the thread's column to lie inside the extent ``name`` is accessed on."""
        (ilo, ihi), (jlo, jhi) = self.plan.field_extents.get(name, analysis.ZERO_EXTENT)
        (silo, sihi), (sjlo, sjhi) = stage.extent
        conds = []
        if ilo > silo:
            conds.append(f"i >= {ilo}")
        if ihi < sihi:
            conds.append(f"i < a.dI + ({ihi})")
        if jlo > sjlo:
            conds.append(f"j >= {jlo}")
        if jhi < sjhi:
            conds.append(f"j < a.dJ + ({jhi})")
        return " && ".join(conds) if conds else None

    def local_decls(self, nest: Nest, indent: str) -> None:
        seen = []
        for s in nest.stmts:
            n = s.target.name
            if n in self.plan.locals and n not in seen:
                seen.append(n)
                self.lines.append(f"{indent}{_CTYPE[self.decl_dtype[n].name]} l_{_c_ident(n)}{self.local_suffix};")

    @staticmethod
    def bound(b: ir.AxisBound) -> str:
        return f"{b.offset}" if b.level is ir.Level.START else f"(a.dK + ({b.offset}))"

    # -- kernels ------------------------------------------------------------------------------
    def prefetch_chunk(self, group: Sequence[Stmt], stage: Stage, nest: Nest, active: Sequence[str], back: int):
        """Which reads of a column loop may be issued a whole chunk of K levels early, and how deep the chunk is.

        Returns (depth, {(name, (di, dj), level relative to the chunk's first level, data index): register},
        {id(statement): [(name, offset, data index), ...]}) or None.  A read is hoistable when nothing the chunk
        stores can be the value it is meant to see: fields the nest does not write; for fields it does write
        (level-by-level, at offset 0 only), the levels AHEAD of the sweep, and the current level for reads that
        come before the first write in statement order.  Statements under a narrower extent, a horizontal region
        or a `while` keep the plain loop (their reads are only inside the arrays where they execute)."""
        depth = int(TUNING["prefetch"])
        iv = nest.interval
        if iv.start.level is iv.end.level:  # statically short interval: no deeper than it is long
            while depth > iv.end.offset - iv.start.offset:
                depth //= 2
        if depth < 2:
            return None
        for s in group:
            if s.extent != stage.extent or s.region is not None or s.loops:
                return None
        written = {s.target.name for s in group}
        displaced = {s.target.name for s in group if s.target.offset != (0, 0, 0) or s.target.koffset is not None
                     or any(isinstance(d, ir.Expr) for d in s.target.data_index or ())}
        ahead = 1 if nest.order is ir.LoopOrder.FORWARD else -1 if nest.order is ir.LoopOrder.BACKWARD else 0
        per_statement: Dict[int, List[Tuple]] = {}
        done: Set[str] = set()  # written by an earlier statement of the level
        for s in group:
            keys = []
            for e in _stmt_field_reads(s):
                name = e.name
                if (e.koffset is not None or name in self.plan.locals or name in self.plan.register_only
                        or any(isinstance(d, ir.Expr) for d in e.data_index or ())):
                    continue
                if (self.tc is not None and self.tc_sweep2 and self.tc_mode[0] != "mem" and name in self.tc[0].names
                        and tuple(e.offset) == (0, 0, 0)):
                    continue  # on chip in this range of the backward sweep: nothing to fetch
                if e.offset[:2] == (0, 0) and name in active and e.offset[2] == back:
                    continue  # served by the forwarded register
                if name in written:
                    dk = e.offset[2]
                    if name in displaced or "K" not in self.axes.get(name, ("I", "J", "K")):
                        continue
                    if not ((ahead and dk * ahead > 0) or (dk == 0 and name not in done)):
                        continue
                key = (name, tuple(e.offset), tuple(e.data_index or ()))
                if key not in keys:
                    keys.append(key)
            per_statement[id(s)] = keys
            done.add(s.target.name)
        if not any(per_statement.values()):
            return None
        while depth >= 2:
            loads: Dict[Tuple, str] = {}
            for u in range(depth):
                step = -u if nest.order is ir.LoopOrder.BACKWARD else u
                for keys in per_statement.values():
                    for name, off, data in keys:
                        slot = (name, off[:2], off[2] + step, data)
                        if slot not in loads:
                            loads[slot] = f"pf{len(loads)}_{_c_ident(name)}"
            if len(loads) <= int(TUNING["prefetch_loads"]):
                return depth, loads, per_statement
            depth //= 2
        return None

    def stage_globals(self, stage: Stage) -> List[str]:
        names = []
        for nest in stage.nests:
            for s in nest.stmts:
                for n in [s.target.name] + [e.name for e in _stmt_field_reads(s)]:
                    if n not in self.plan.locals and n not in names:
                        names.append(n)
        return names

    def _kernel_header(self, stage: Stage, kname: str, block, j_per_thread: int) -> None:
        """Signature, tile -> (i, j), base pointers of the fields the stage touches."""
        L = self.lines
        (ilo, ihi), (jlo, jhi) = stage.extent
        L.append(f'extern "C" __global__ void __launch_bounds__({block[0] * block[1]}) {kname}(const gt_args a) {{')
        L.append("    unsigned gt_bx, gt_by, gt_bz;")
        L.append(f"    gt_tile({TUNING['xcd_rows']}u, gt_bx, gt_by, gt_bz);")
        L.append(f"    const gt_i64 i = (gt_i64)gt_bx * {block[0]} + threadIdx.x + ({ilo});")
        L.append(f"    if (i >= a.dI + ({ihi})) return;")
        if j_per_thread > 1:
            L.append("    #pragma unroll")
            L.append(f"    for (int jv = 0; jv < {j_per_thread}; ++jv) {{")
            L.append(f"    const gt_i64 j = ((gt_i64)gt_by * {block[1]} + threadIdx.y) * {j_per_thread} + jv + ({jlo});")
            L.append(f"    if (j >= a.dJ + ({jhi})) break;")
        else:
            L.append(f"    const gt_i64 j = (gt_i64)gt_by * {block[1]} + threadIdx.y + ({jlo});")
            L.append(f"    if (j >= a.dJ + ({jhi})) return;")
        for n in self.stage_globals(stage):
            if n in self.plan.register_only:
                continue
            c = _c_ident(n)
            ct = _CTYPE[self.decl_dtype[n].name]
            axes = self.axes.get(n, ("I", "J", "K"))
            const = "" if n in stage.written else "const "
            # scratch never aliases anything; API arrays only when the host checked that they are disjoint
            qual = " __restrict__" if n in self.plan.scratch else " GT_RESTRICT"
            off = " + ".join(t for t in (f"i * GT_SI(a.{c}_si)" if "I" in axes else "", f"j * a.{c}_sj" if "J" in axes else "") if t) or "0"
            L.append(f"    {const}{ct}* const{qual} b_{c} = a.{c} + {off};")

    def _column_range(self, si: int, stage: Stage, nest: Nest, group: Sequence[Stmt], active: Sequence[str], back: int,
                      lo: str, hi: str) -> None:
        """The K loop of one statement group of a column nest over the levels [lo, hi) (C expressions), in sweep
        order: chunks of levels whose loads are issued ahead where that is provably safe, else level by level."""
        L = self.lines
        backward = nest.order is ir.LoopOrder.BACKWARD
        stage_fwd = {n: d for (s_i, n), d in self.plan.forwarded.items() if s_i == si}

        def level(kexpr: str, pad: str) -> None:
            self._emit_level(si, stage, nest, group, active, back, kexpr, pad)

        loop = (f"for (gt_i64 k = ({hi}) - 1; k >= ({lo}); --k)" if backward else f"for (gt_i64 k = ({lo}); k < ({hi}); ++k)")

        def plain_loop() -> None:
            if TUNING["unroll"] > 1:
                L.append(f"        #pragma unroll {TUNING['unroll']}")
            L.append(f"        {loop} {{")
            level("k", "            ")
            L.append("        }")

        chunk = None if nest.split_statements else self.prefetch_chunk(group, stage, nest, active, back)
        if chunk is None:
            plain_loop()
            return
        # Loads of `depth` levels are issued before the arithmetic that depends on them.  Only valid when
        # no store of the chunk can alias a hoisted load: the host has proven the arrays disjoint
        # (GT4MI_NO_ALIAS), and within one array the levels of a chunk are different addresses.
        depth, loads, per_statement = chunk
        L.append("#if GT4MI_NO_ALIAS")
        L.append("        {")
        L.append(f"        gt_i64 k = {'(' + hi + ') - 1' if backward else '(' + lo + ')'};")
        L.append(f"        for (; {'k - ' + str(depth - 1) + ' >= (' + lo + ')' if backward else 'k + ' + str(depth) + ' <= (' + hi + ')'}; "
                 f"k {'-' if backward else '+'}= {depth}) {{")
        self.tc_written = set()  # hoisted loads see what memory held BEFORE the chunk's levels are assigned
        for (name, off, rel, data), var in loads.items():
            e = ir.FieldAccess(name, (off[0], off[1], rel), None, None, data)
            L.append(f"            const {_CTYPE[self.decl_dtype[name].name]} {var} = {self.access(e, 'k', -1, {})};")
        for u in range(depth):
            step = -u if backward else u
            self.prefetch_for = {sid: {key: loads[(key[0], key[1][:2], key[1][2] + step, key[2])] for key in keys}
                                 for sid, keys in per_statement.items()}
            L.append("            {")
            level(f"(k {'-' if backward else '+'} {u})", "                ")
            L.append("            }")
        self.prefetch_for = {}
        L.append("        }")
        L.append(f"        for (; {'k >= (' + lo + '); --k' if backward else 'k < (' + hi + '); ++k'}) {{")
        level("k", "            ")
        L.append("        }")
        L.append("        }")
        L.append("#else")
        plain_loop()
        L.append("#endif")

    def _column_body(self, si: int, stage: Stage) -> None:
        """Thread per column: the nests of the stage one after the other, K serial per thread.  With ``self.tc`` set
        (the `_tc` variant of a two-sweep stage) every nest is cut into up to three ranges of levels -- memory, LDS,
        registers -- and the cached fields are stored / read where the level lives (stage_planner.TopCache)."""
        L = self.lines
        stage_fwd = {n: d for (s_i, n), d in self.plan.forwarded.items() if s_i == si}
        for n in stage_fwd:  # r_<n>: the level behind the sweep, alive across the stage's nests
            L.append(f"    {_CTYPE[self.decl_dtype[n].name]} r_{_c_ident(n)} = {_CTYPE[self.decl_dtype[n].name]}();")
        if self.tc is not None:
            cache, n_reg, n_lds, threads = self.tc
            L.append(f"    const unsigned tc_tid = threadIdx.y * {threads // max(1, TUNING['block_column'][1])} + threadIdx.x;")
            for name in cache.names:
                c, ct = _c_ident(name), _CTYPE[self.decl_dtype[name].name]
                if n_lds:
                    L.append(f"    __shared__ {ct} tc_lds_{c}[{n_lds * threads}];")
                if n_reg:  # every slot is written by the forward sweep before the backward sweep reads it
                    L.append(f"    {ct} {', '.join(f'tc_{c}_{u}' for u in range(n_reg))};")
        for ni, nest in enumerate(stage.nests):
            self.nt_nest_order = nest.order
            if self.tc is not None and ni == self.tc[0].first_sweep_nests:
                self._second_sweep_bases(stage)
            L.append("    {")
            L.append(f"        const gt_i64 k0 = {self.bound(nest.interval.start)}, k1 = {self.bound(nest.interval.end)};")
            back = -1 if nest.order is ir.LoopOrder.FORWARD else 1
            active = [n for n, d in stage_fwd.items() if nest.order is not ir.LoopOrder.PARALLEL and d == back
                      and any(s.target.name == n or any(e.name == n for e in _stmt_field_reads(s)) for s in nest.stmts)]
            groups = [[s] for s in nest.stmts] if nest.split_statements else [nest.stmts]
            first = "k0" if nest.order is ir.LoopOrder.FORWARD else "(k1 - 1)"
            for n in active:
                mode, prev = self.plan.prime.get((si, ni, n), (None, None))
                if mode in (None, "carried"):
                    continue
                # Priming from memory: the first iteration's read of the level behind is an access the
                # stencil makes anyway, so it is inside the array whenever the loop runs at all and the
                # column lies in the extent the field is accessed on.
                conds = ["k1 > k0"]
                if mode == "prime_if_prev_empty":
                    piv = stage.nests[prev].interval
                    conds.append(f"!({self.bound(piv.end)} > {self.bound(piv.start)})")
                cond = self.column_in_extent(n, stage)
                if cond:
                    conds.append(cond)
                L.append(f"        if ({' && '.join(conds)}) r_{_c_ident(n)} = "
                         f"{self.access(ir.FieldAccess(n, (0, 0, back)), first, -1, {})};")
            for group in groups:
                if self.tc is None:
                    self._column_range(si, stage, nest, group, active, back, "k0", "k1")
                    continue
                cache, n_reg, n_lds, threads = self.tc
                self.tc_sweep2 = ni >= cache.first_sweep_nests
                iv = nest.interval
                if iv.end.level is ir.Level.START:  # ends below the cached range (the host guarantees it)
                    self.tc_mode = ("mem",)
                    self._column_range(si, stage, nest, group, active, back, "k0", "k1")
                    continue
                # relative to dK: the nest covers [lo_rel (None: somewhere below the cache), hi_rel)
                hi_rel = iv.end.offset
                lo_rel = iv.start.offset if iv.start.level is ir.Level.END else None
                ranges = []  # in ascending order of levels
                if lo_rel is None or lo_rel < -(n_reg + n_lds):
                    ranges.append((("mem",), "k0", f"gt_min(k1, a.dK - {n_reg + n_lds})"))
                first_slot = 0
                if n_lds and TUNING["top_cache_unroll_lds"]:
                    first_slot = -n_lds  # the LDS levels are straight-line code too: slots -n_lds .. -1
                elif n_lds and hi_rel > -(n_reg + n_lds) and (lo_rel is None or lo_rel < -n_reg):
                    ranges.append((("lds",), f"gt_max(k0, a.dK - {n_reg + n_lds})", f"gt_min(k1, a.dK - {n_reg})"))
                slots = [u for u in range(first_slot, n_reg) if -n_reg + u < hi_rel and (lo_rel is None or -n_reg + u >= lo_rel)]
                if slots:
                    ranges.append((("reg", slots), None, None))
                if nest.order is ir.LoopOrder.BACKWARD:
                    ranges.reverse()
                for mode, lo, hi in ranges:
                    if mode[0] != "reg":
                        self.tc_mode = mode
                        L.append("        {")
                        self._column_range(si, stage, nest, group, active, back, lo, hi)
                        L.append("        }")
                        continue
                    self._register_range(si, stage, nest, group, active, back, mode[1], n_reg)
                self.tc_mode = ("mem",)
            L.append("    }")
        if self.tc is not None:
            L.append("    }")  # the scope _second_sweep_bases opened

    def _second_sweep_bases(self, stage: Stage) -> None:
        """The second sweep of a `_tc` kernel addresses its arrays through base pointers the optimiser cannot relate to
        the first sweep's (base + an opaque zero).  A field both sweeps touch at the same levels (u_pos and
        utens_stage of the vertical advection) otherwise has every register-range address of the first sweep kept
        alive -- in accumulator registers -- for its reuse by the second: 2 registers per field and level, as much
        as the cached values themselves."""
        L = self.lines
        names = [n for n in self.stage_globals(stage) if n not in self.plan.register_only]
        L.append("    gt_i64 tc_zero = 0;")
        L.append('    asm volatile("" : "+s"(tc_zero));')
        for n in names:
            c = _c_ident(n)
            L.append(f"    const auto tc_b_{c} = b_{c} + tc_zero;")
        L.append("    {")
        for n in names:
            c = _c_ident(n)
            ct = _CTYPE[self.decl_dtype[n].name]
            const = "" if n in stage.written else "const "
            qual = " __restrict__" if n in self.plan.scratch else " GT_RESTRICT"
            L.append(f"    {const}{ct}* const{qual} b_{c} = tc_b_{c};")

    def _emit_level(self, si: int, stage: Stage, nest: Nest, group, active, back: int, kexpr: str, pad: str) -> None:
        """One K level of a column sweep: locals, statements, rotation of the forwarded registers."""
        L = self.lines
        self.tc_written = set()
        self.local_decls(Nest(nest.order, nest.interval, group, nest.block_id), pad)
        carry = [n for n in active if any(s.target.name == n for s in group)]
        reg: Dict[Tuple[str, int], str] = {(n, back): f"r_{_c_ident(n)}" for n in active}
        for n in carry:
            L.append(f"{pad}{_CTYPE[self.decl_dtype[n].name]} n_{_c_ident(n)} = r_{_c_ident(n)};")
        self.statements(group, stage, si, kexpr, reg, pad, carry)
        for n in active:
            c = _c_ident(n)
            if n in carry:
                L.append(f"{pad}r_{c} = n_{c};")
            else:  # only read in this nest: rotate in the level just passed
                cond = self.column_in_extent(n, stage)
                load = self.access(ir.FieldAccess(n, (0, 0, 0)), kexpr, -1, {})
                L.append(f"{pad}{'if (' + cond + ') ' if cond else ''}r_{c} = {load};")

    def _register_range(self, si: int, stage: Stage, nest: Nest, group, active, back: int, slots: Sequence[int], n_reg: int) -> None:
        """The levels of a nest that live in register slots (compile-time indices): straight-line code, in sweep
        order, in batches whose loads are issued ahead of the arithmetic exactly as in the loops (prefetch_chunk) -- and
        one batch further: the loads of batch n + 1 are issued before the arithmetic of batch n, so that a lone wave per
        SIMD always has a batch of loads in flight while it computes (hoisting them over the stores of batch n is safe
        for the same reason hoisting within a batch is: prefetch_chunk only hoists reads of levels the sweep has not
        stored yet)."""
        L = self.lines
        backward = nest.order is ir.LoopOrder.BACKWARD
        order = list(reversed(slots)) if backward else list(slots)
        self.tc_mode = ("reg", order[0])
        chunk = None if nest.split_statements else self.prefetch_chunk(group, stage, nest, active, back)
        depth = chunk[0] if chunk is not None else 1
        batches = [order[pos:pos + depth] for pos in range(0, len(order), depth)]
        sign = "-" if backward else "+"

        def base(batch) -> str:
            return f"(a.dK - {n_reg - batch[0]})"

        def full(batch) -> bool:
            return chunk is not None and len(batch) == depth

        nest_writes = {s.target.name for s in nest.stmts}
        loaded: Dict[Tuple, str] = {}  # (field, (di, dj), slot of the level, data index) -> register that holds it

        def issue_loads(bi: int) -> Dict[Tuple, str]:
            """Declare and issue the hoisted loads of batch `bi` (names unique per batch: one scope holds them all).
            A value an earlier batch of this range already fetched (wcon[k+1] of one batch is wcon[k] of the next)
            is taken from its register when nothing in the nest writes the field."""
            _, loads, _ = chunk
            named = {}
            self.tc_written = set()  # hoisted loads see what memory held BEFORE the batch's levels are assigned
            for (name, off, rel, data), var in loads.items():
                where = (name, off, batches[bi][0] + rel, data)  # slots count levels from a.dK - n_reg, in K order
                if name not in nest_writes and where in loaded:
                    named[(name, off, rel, data)] = loaded[where]
                    continue
                e = ir.FieldAccess(name, (off[0], off[1], rel), None, None, data)
                L.append(f"            const {_CTYPE[self.decl_dtype[name].name]} q{bi}_{var} = {self.access(e, base(batches[bi]), -1, {})};")
                named[(name, off, rel, data)] = loaded[where] = f"q{bi}_{var}"
            return named

        if chunk is not None and TUNING["top_cache_pipeline"] == 2:
            self._register_range_rolling(si, stage, nest, group, active, back, order, n_reg, chunk, nest_writes)
            return
        L.append("        {")
        pending = issue_loads(0) if full(batches[0]) and TUNING["top_cache_pipeline"] else None
        for bi, batch in enumerate(batches):
            # a fence per batch: otherwise the scheduler hoists the loads of ALL unrolled levels to the top of the
            # straight-line region and runs out of registers
            L.append("        __builtin_amdgcn_sched_barrier(0);")
            named, pending = pending, None
            if not TUNING["top_cache_pipeline"]:
                named = issue_loads(bi) if full(batch) else None
            elif bi + 1 < len(batches) and full(batches[bi + 1]):
                pending = issue_loads(bi + 1)
            if named is not None:
                _, _, per_statement = chunk
                for u, slot in enumerate(batch):
                    step = -u if backward else u
                    self.prefetch_for = {sid: {key: named[(key[0], key[1][:2], key[1][2] + step, key[2])] for key in keys}
                                         for sid, keys in per_statement.items()}
                    self.tc_mode = ("reg", slot)
                    L.append("            {")
                    self._emit_level(si, stage, nest, group, active, back, f"({base(batch)} {sign} {u})", "                ")
                    L.append("            }")
                    self._pin_register_level(slot)
                self.prefetch_for = {}
            else:  # the levels that do not fill a batch: one at a time
                for u, slot in enumerate(batch):
                    self.tc_mode = ("reg", slot)
                    L.append("            {")
                    self._emit_level(si, stage, nest, group, active, back, f"({base(batch)} {sign} {u})", "                ")
                    L.append("            }")
                    self._pin_register_level(slot)
        L.append("        }")
        L.append("        __builtin_amdgcn_sched_barrier(0);")

    def _register_range_rolling(self, si: int, stage: Stage, nest: Nest, group, active, back: int, order: Sequence[int],
                                n_reg: int, chunk, nest_writes: Set[str]) -> None:
        """Register levels with a rolling prefetch: level n issues the loads of level n + D (D = the chunk depth of
        prefetch_chunk) and then computes itself, so D levels of loads are always in flight and every level's arithmetic
        overlaps with them -- the registers of one more level instead of a second batch buffer.  Values the sweep
        already holds (wcon[k + 1] of one level is wcon[k] of the next) are not fetched again."""
        L = self.lines
        depth, _, per_statement = chunk
        if TUNING["top_cache_lookahead"] > 0:
            depth = int(TUNING["top_cache_lookahead"])
        keys = []  # (field, (di, dj, dk), data index) read through hoisted loads, in statement order
        for ks in per_statement.values():
            for key in ks:
                if key not in keys:
                    keys.append(key)
        loaded: Dict[Tuple, str] = {}

        def kexpr(slot: int) -> str:
            return f"(a.dK - {n_reg - slot})"

        def issue(slot: int) -> None:
            self.tc_written = set()  # hoisted loads see what memory held BEFORE the levels in between are assigned
            for name, off, data in keys:
                where = (name, tuple(off[:2]), slot + off[2], data)
                if where in loaded and name not in nest_writes:
                    continue
                var = f"q{len(loaded)}_{_c_ident(name)}"
                e = ir.FieldAccess(name, tuple(off), None, None, data)
                L.append(f"            const {_CTYPE[self.decl_dtype[name].name]} {var} = {self.access(e, kexpr(slot), -1, {})};")
                loaded[where] = var

        L.append("        {")
        for slot in order[:depth]:
            issue(slot)
        for n, slot in enumerate(order):
            L.append("        __builtin_amdgcn_sched_barrier(0);")
            if n + depth < len(order):
                issue(order[n + depth])
            self.prefetch_for = {sid: {key: loaded[(key[0], tuple(key[1][:2]), slot + key[1][2], key[2])] for key in ks}
                                 for sid, ks in per_statement.items()}
            self.tc_mode = ("reg", slot)
            L.append("            {")
            self._emit_level(si, stage, nest, group, active, back, kexpr(slot), "                ")
            L.append("            }")
            self._pin_register_level(slot)
        self.prefetch_for = {}
        L.append("        }")
        L.append("        __builtin_amdgcn_sched_barrier(0);")

    def _pin_register_level(self, slot: int) -> None:
        """A level whose results only go to register slots has no side effect, and instruction selection sinks its
        arithmetic to the first use of those slots -- the other sweep -- across every sched_barrier in between (they
        order the scheduler, not the selector).  All loads of all register levels would then stay live until the second
        sweep starts (measured: 12 registers per cached level instead of 4, spills from 24 levels on).  An empty
        volatile asm that takes the slots as inputs pins the arithmetic to the level it belongs to."""
        if self.tc_sweep2 or not self.tc_written or slot < 0:  # (an LDS level's store is its own anchor)
            return
        names = sorted(self.tc_written)
        constraints = ", ".join(f'"v"(tc_{_c_ident(n)}_{slot})' for n in names)
        self.lines.append(f"            asm volatile(\"\" :: {constraints});")

    def _nt_load_fields(self, stage: Stage) -> Optional[Set[str]]:
        """Fields of a column stage whose loads from memory are nontemporal (TUNING["column_nt_loads"])."""
        mode = int(TUNING["column_nt_loads"])
        if mode <= 0:
            return None
        names, shifted, orders = set(), set(), {}
        cached = set(self.tc[0].names) if self.tc is not None else set()
        for nest in stage.nests:
            for st in nest.stmts:
                for ex in ([c for _, c in st.loops] + ([st.value] if st.mask is None else [st.mask, st.value]) + stage_planner_target_exprs(st.target)):
                    for e in ir.walk(ex):
                        if isinstance(e, ir.FieldAccess) and e.name not in self.plan.locals:
                            names.add(e.name)
                            if tuple(e.offset[:2]) != (0, 0):
                                shifted.add(e.name)
                            if not (e.name in cached and nest.order is ir.LoopOrder.BACKWARD):  # (served on chip in the second sweep)
                                orders.setdefault(e.name, set()).add(nest.order)
        ok = {n for n in names if n in self.decl_dtype and self.decl_dtype[n].kind in "fiu" and self.decl_dtype[n].itemsize >= 4}
        if mode == 1:
            return ok
        ok -= shifted
        if mode == 2:
            return ok
        both = {n for n in ok if len(orders.get(n, ())) > 1 and n not in stage.written}
        ok = {n for n in ok if len(orders.get(n, ())) <= 1}  # 3: ... and read from memory by ONE sweep only
        if mode == 3:
            return ok
        if mode == 5:  # 3 + what BOTH sweeps read (and nobody writes): cacheable in the first sweep, nontemporal in the last
            last = [n.order for n in stage.nests][-1]
            self.nt_loads_last = {n: last for n in both}
            return ok
        return ok - set(stage.written)  # 4: ... and never written by the stage

    def kernel(self, si: int, stage: Stage, kname: str) -> KernelSource:
        L = self.lines
        if stage.mapping == "ijk":
            bi, bj, k_per_thread, j_per_thread = (tuple(TUNING["block_ijk"]) + (1,))[:4]
        else:
            (bi, bj), k_per_thread, j_per_thread = TUNING["block_column"], 1, 1
        block = (bi, bj, 1)
        self._kernel_header(stage, kname, block, j_per_thread)
        if stage.mapping == "ijk":
            if k_per_thread > 1:
                L.append(f"    #pragma unroll")
                L.append(f"    for (int kk = 0; kk < {k_per_thread}; ++kk) {{")
                L.append(f"    const gt_i64 k = a.k_lo + (gt_i64)gt_bz * {k_per_thread} + kk;")
            else:
                L.append("    const gt_i64 k = a.k_lo + gt_bz;")
            for nest in stage.nests:
                L.append(f"    if (k >= {self.bound(nest.interval.start)} && k < {self.bound(nest.interval.end)} && k < a.k_hi) {{")
                self.local_decls(nest, "        ")
                self.statements(nest.stmts, stage, si, "k", {}, "        ")
                L.append("    }")
            if k_per_thread > 1:
                L.append("    }")
        else:
            self.nt_loads = self._nt_load_fields(stage)
            try:
                self._column_body(si, stage)
            finally:
                self.nt_loads, self.nt_loads_last, self.nt_nest_order = None, {}, None
        if j_per_thread > 1:
            L.append("    }")
        L.append("}")
        L.append("")
        top_cache = None
        cache = self.plan.top_cache.get(si) if stage.mapping == "column" else None
        n_reg_cfg, lds_bytes, lds_cap = (tuple(TUNING["top_cache"]) + (0, 0, 64)[len(TUNING["top_cache"]):])[:3]
        if cache is not None and (n_reg_cfg != 0 or lds_bytes > 0):
            threads = block[0] * block[1]
            per_level = sum(self.decl_dtype[n].itemsize for n in cache.names) * threads
            n_lds = min(int(lds_bytes) // per_level, int(lds_cap))
            if n_reg_cfg >= 0:
                depths = [int(n_reg_cfg)]
            else:
                budget, step = (tuple(TUNING["top_cache_auto"]) + (448, 24)[len(TUNING["top_cache_auto"]):])[:2]
                n_max = min(int(budget) // sum(self.decl_dtype[n].itemsize // 4 for n in cache.names), 128)
                n_max -= n_max % 8
                # the deepest one is the likeliest to spill (the host then takes the next): a close second, then coarse steps
                depths = [n_max] + list(range(n_max - 8, 16, -max(8, int(step)))) + [min(16, n_max)]
                depths = sorted({d for d in depths if d > 0}, reverse=True)
            variants = []
            for n_reg in depths:
                if n_reg + n_lds <= 0:
                    continue
                # every START-relative interval bound must lie below the cached range, and every nest that is not
                # statically empty must hold at least one level there -> the smallest domain the variant may run on
                variants.append((int(n_reg), int(n_lds), int(n_reg) + int(n_lds) + cache.start_margin + 1))
                L.append("#if GT4MI_NO_ALIAS")  # the cached copies stand in for memory: only with disjoint arguments
                self._kernel_header(stage, f"{kname}_tc{n_reg}", block, 1)
                self.tc = (cache, int(n_reg), int(n_lds), threads)
                self.nt_loads = self._nt_load_fields(stage)
                try:
                    self._column_body(si, stage)
                finally:
                    self.tc, self.tc_mode, self.tc_sweep2, self.tc_written, self.nt_loads = None, ("mem",), False, set(), None
                    self.nt_loads_last, self.nt_nest_order = {}, None
                L.append("}")
                L.append("#endif")
                L.append("")
            top_cache = tuple(variants) or None
        vec = _vector_width(self, stage) if j_per_thread == 1 and block[0] % 64 == 0 else 0
        vec_fields: Tuple[str, ...] = ()
        vec_rows, xcd_rows, vec_block = _strip_shape(self, stage, block) if vec else (max(1, TUNING["vector_rows"]), TUNING["xcd_rows"], block)
        if vec:
            vec_fields = _emit_vector_kernel(self, si, stage, kname, vec, vec_rows, vec_block, k_per_thread, xcd_rows)
        shared_halo, shared_vec, shared_fields, shared_rows, shared_preferred = 0, 0, (), 0, False
        svec = vec or (_vector_width(self, stage, any_reach=True) if j_per_thread == 1 and block[0] % 64 == 0 else 0)
        if svec:
            form = _shared_form(self, stage, svec, k_per_thread)
            if form is not None:
                shared_vec = svec
                shared_rows = int(TUNING["shared_rows"]) or (8 if svec >= 4 else 4)
                shared_halo, shared_fields = _emit_shared_kernel(self, si, stage, kname, svec, shared_rows, block, form)
                shared_preferred = not vec or any(np.dtype(defs[v].dtype).itemsize == 8 for _, _, defs, need in form[0] for v in need)
        plane = None if stage.plane is None else (stage.plane[0], stage.plane[1].value, stage.plane[2])
        return KernelSource(kname, stage.mapping, stage.extent, block, k_per_thread, j_per_thread, vec, vec_fields,
                            vec_rows if vec else 1, plane, top_cache, shared_halo, shared_rows if shared_halo else 0,
                            shared_vec, shared_fields, shared_preferred, vec_block if vec and vec_block != block else None)


def _vector_width(em: "_Emitter", stage: Stage, any_reach: bool = False) -> int:
    """How many consecutive I points a lane may own in this stage (0 = keep one point per thread).  ``any_reach``: do
    not insist that every (inlined) read stays within one lane of the point -- the `_vecs` form reaches further through
    its temporaries, one lane per link of the chain.

    Needs: thread-per-point mapping, the stage starting at the domain's first column, reads only of arrays
    the stage does not write (so rows can be loaded once, up front), every statement on the full stage
    extent and outside horizontal regions, I offsets within one lane's reach, 4- or 8-byte elements."""
    if stage.mapping != "ijk" or stage.extent[0][0] != 0 or not TUNING["vector"]:
        return 0
    sizes = set()
    for nest in stage.nests:
        for s in nest.stmts:
            # the strip form loads every row a nest reads before any statement runs, unconditionally: that is
            # only inside the arrays when each statement covers the whole stage extent and none is restricted
            # to a horizontal region (whose reads need less halo than an unrestricted statement's would)
            if s.extent != stage.extent or s.region is not None or s.loops:
                return 0
            if s.target.offset != (0, 0, 0) or s.target.data_index or s.target.koffset is not None:
                return 0
            if s.target.name not in em.plan.locals:
                if "I" not in em.axes.get(s.target.name, ("I", "J", "K")):
                    return 0
                sizes.add(em.decl_dtype[s.target.name].itemsize)
            for e in _stmt_field_reads(s):
                if e.koffset is not None or e.data_index:
                    return 0
                if e.name in em.plan.locals:
                    continue
                if e.name in stage.written:
                    return 0
                if "I" in em.axes.get(e.name, ("I", "J", "K")):
                    sizes.add(em.decl_dtype[e.name].itemsize)
    if not sizes or not sizes <= {4, 8}:
        return 0
    vec = 16 // max(sizes)
    if any_reach:
        return vec
    for nest in stage.nests:
        for s in nest.stmts:
            if any(abs(e.offset[0]) > vec for e in _stmt_field_reads(s) if e.name not in em.plan.locals):
                return 0
    return vec


def _strip_shape(em: "_Emitter", stage: Stage, block=(64, 4, 1)) -> Tuple[int, int, Tuple[int, int, int]]:
    """(J rows per lane, rows per XCD run, workgroup shape) of a stage's strip kernel.  Measured in round 1
    (profiles/r1_codegen_sweep.log, r1_codegen_xcd_rows_strip_kernels.log): 8 rows per lane and XCD runs of 4 tile rows
    help the Laplacian (+2 % and +4 %) and hurt horizontal diffusion (-8 % registers, -5 %), so the tall, XCD-grouped shape
    is for LIGHT stages only: one statement reading one 8-byte array within one row / column of the point.  Round 3: those
    stages also get the hand-written kernel's workgroup, 256 lanes along I x 1 (a tile of 512 columns x 8 rows instead of
    128 x 32: the wave edges' neighbour columns are then mostly inside the workgroup) -- 364 -> 368 GLUPS on the 512^3
    Laplacian (profiles/r3_codegen_strip_blocks.log).  Explicit settings of GT4MI_CODEGEN_VECTOR_ROWS / _XCD_ROWS /
    _BLOCK_IJK win."""
    import os

    rows, xcd = max(1, TUNING["vector_rows"]), TUNING["xcd_rows"]
    if "GT4MI_CODEGEN_VECTOR_ROWS" in os.environ or "GT4MI_CODEGEN_XCD_ROWS" in os.environ:
        return rows, xcd, tuple(block)
    stmts = [s for nest in stage.nests for s in nest.stmts]
    reads = [e for s in stmts for e in _stmt_field_reads(s) if e.name not in em.plan.locals]
    arrays = {e.name for e in reads}
    light = (len(stmts) == 1 and len(stage.nests) == 1 and len(arrays) == 1
             and all(max(abs(e.offset[0]), abs(e.offset[1])) <= 1 and e.offset[2] == 0 for e in reads)
             and all(em.decl_dtype[n].itemsize == 8 for n in arrays | {stmts[0].target.name}))
    if not light:
        return rows, xcd, tuple(block)
    return 8, 4, (tuple(block) if "GT4MI_CODEGEN_BLOCK_IJK" in os.environ else (256, 1, 1))


def _shared_form(em: "_Emitter", stage: Stage, vec: int, k_per_thread: int):
    """([(nest, order, defs, rows needed per version), ...], halo lanes) when the stage may get a `_vecs` kernel -- a
    strip kernel whose inlined temporaries are computed ONCE per point and handed to the neighbouring lanes --, else
    None.  Eligible: PARALLEL interval blocks the stage holds completely, at least one of them with inlined temporaries;
    plain (unmasked, unshifted) assignments to arrays nobody in the stage reads; every operand a full I, J, K array read
    at a compile-time offset of at most `vec` points in I."""
    if not TUNING["shared_temporaries"] or vec <= 0 or k_per_thread != 1 or not stage.nests:
        return None
    if stage.extent != analysis.ZERO_EXTENT:
        return None
    stage_written = {s.target.name for nest in stage.nests for s in nest.stmts}
    forms, width = [], 0
    for nest in stage.nests:
        if nest.order is not ir.LoopOrder.PARALLEL or nest.split_statements:
            return None
        one = _shared_nest_form(em, nest, vec, stage_written)
        if one is None:
            return None
        forms.append((nest,) + one[:3])
        width = max(width, one[3])
    if width == 0 or not any(need for _, _, _, need in forms):
        return None  # nothing to pass between lanes
    halo = -(-width // vec)
    if halo > 4:
        return None
    return forms, halo


def _shared_nest_form(em: "_Emitter", nest: Nest, vec: int, stage_written: Set[str]):
    """(order, defs, need, I reach of the inputs) of one nest of a `_vecs` kernel, or None (see _shared_form)."""
    form = em.plan.shared_forms.get(nest.block_id)
    if form is None:  # a block without inlined temporaries (a boundary interval, say): its statements as they are
        form = ([("stmt", s) for s in nest.stmts], {})
    order, defs = form
    # temporaries of the block that were NOT inlined (read at their own point only) become versions as well
    current: Dict[str, str] = {}

    def renamed(expr: ir.Expr) -> ir.Expr:
        def fn(e):
            if isinstance(e, ir.FieldAccess) and e.name in current:
                return _replace(e, name=current[e.name])
            return e

        return ir.map_expr(expr, fn)

    new_order: List[Tuple[str, object]] = []
    new_defs: Dict[str, ir.Expr] = {}
    for kind, obj in order:
        if kind == "def":
            new_defs[obj] = renamed(defs[obj])
            new_order.append(("def", obj))
        elif obj.target.name in em.global_names and obj.target.name not in em.plan.locals:
            if obj.region is not None or obj.loops:
                return None
            # (a run-time `if` around an assignment to an array: stored point by point where the condition holds)
            new_order.append(("stmt", _replace(obj, value=renamed(obj.value), mask=renamed(obj.mask) if obj.mask is not None else None)))
        else:  # a thread-local temporary
            t = obj.target
            if obj.region is not None or obj.loops or tuple(t.offset) != (0, 0, 0) or t.koffset is not None or t.data_index:
                return None
            version = f"{t.name}__w{len(new_defs)}"
            value = renamed(obj.value)
            if obj.mask is not None:  # where(mask, value, what it held): a select, as in the planner's substitution
                dt = np.dtype(em.decl_dtype[t.name])
                old = (ir.FieldAccess(current[t.name], (0, 0, 0), dt) if t.name in current
                       else ir.Literal(False if dt == np.dtype("bool") else dt.type(0).item(), dt))
                value = ir.TernaryOp(renamed(obj.mask), value, old, dt)
            new_defs[version] = value
            new_order.append(("def", version))
            current[t.name] = version
    order, defs = new_order, new_defs
    stmts = [obj for kind, obj in order if kind == "stmt"]
    if sorted(s.target.name for s in nest.stmts if s.target.name not in em.plan.locals) != sorted(st.target.name for st in stmts):
        return None  # the block was cut into several stages
    full = ("I", "J", "K")
    written = {st.target.name for st in stmts} | set(stage_written)
    for st in stmts:
        t = st.target
        if (st.region is not None or st.loops or tuple(t.offset) != (0, 0, 0) or t.koffset is not None
                or t.data_index or t.name not in em.global_names or tuple(em.axes.get(t.name, full)) != full):
            return None
    exprs = [(None, st.value) for st in stmts] + [(None, st.mask) for st in stmts if st.mask is not None] + list(defs.items())
    for _, expr in exprs:
        for e in ir.walk(expr):
            if not isinstance(e, ir.FieldAccess):
                continue
            if e.koffset is not None or e.data_index or abs(e.offset[0]) > vec:
                return None
            if e.name in defs:
                if e.offset[2] != 0:
                    return None
                if e.offset[0] != 0 and np.dtype(defs[e.name].dtype).itemsize not in (4, 8):
                    return None  # (a boolean that crosses lanes: the DPP shifts move 4- or 8-byte values)
            elif (e.name in written or e.name not in em.global_names or e.name in em.plan.register_only
                  or tuple(em.axes.get(e.name, full)) != full):
                return None
    # rows (J extent) every version is needed on, and the reach of the inputs, from the outputs backwards
    need: Dict[str, List[int]] = {}  # version -> [ilo, ihi, jlo, jhi] relative to the output point
    reach = [0, 0]

    def visit(expr: ir.Expr, ext) -> None:
        for e in ir.walk(expr):
            if isinstance(e, ir.FieldAccess):
                shifted = [ext[0] + e.offset[0], ext[1] + e.offset[0], ext[2] + e.offset[1], ext[3] + e.offset[1]]
                if e.name in defs:
                    cur = need.setdefault(e.name, list(shifted))
                    cur[0], cur[1] = min(cur[0], shifted[0]), max(cur[1], shifted[1])
                    cur[2], cur[3] = min(cur[2], shifted[2]), max(cur[3], shifted[3])
                else:
                    reach[0], reach[1] = min(reach[0], shifted[0]), max(reach[1], shifted[1])

    for kind, obj in reversed(order):
        if kind == "stmt":
            visit(obj.value, [0, 0, 0, 0])
            if obj.mask is not None:
                visit(obj.mask, [0, 0, 0, 0])
        elif obj in need:
            visit(defs[obj], need[obj])
    # the halo lanes must hold every value a stored point depends on: the inputs AND the temporaries, which may sit further
    # out than any input (t2[-1] <- t1[-1] <- t0[-1] <- in[+1]: the input two columns away, t0 three -- found by the fuzzer)
    lo = min([reach[0]] + [n[0] for n in need.values()])
    hi = max([reach[1]] + [n[1] for n in need.values()])
    return order, defs, need, max(-lo, hi)


def _emit_shared_kernel(em: "_Emitter", si: int, stage: Stage, kname: str, vec: int, rows_per_lane: int, block, form):
    """``<kname>_vecs``: the strip kernel of a stage with inlined temporaries, with every temporary computed once per
    point.  A lane owns ``vec`` consecutive I points times ``rows_per_lane`` J rows and evaluates each temporary (in the
    single-assignment form the planner kept, stage_planner.inline_horizontal_temporaries_with_forms) at its OWN columns
    only, on the rows its consumers need; a read at an I offset that leaves the lane's columns is a DPP shift of the
    neighbouring lane's value.  The first and last `halo` lanes of a wave are halo lanes: they load and compute like
    every other lane but store nothing (some of what they compute is garbage: the lane beyond the wave does not exist),
    and consecutive waves overlap by 2 * halo lanes -- the design of the hand-written hdiff_jmarch_kernel.  Returns
    (halo, arrays it touches)."""
    forms, halo = form
    L = em.lines
    plan = em.plan
    JT = rows_per_lane
    out_lanes = 64 - 2 * halo
    globals_ = [n for n in em.stage_globals(stage) if n not in plan.register_only]

    L.append(f'extern "C" __global__ void __launch_bounds__({block[0] * block[1]}) {kname}_vecs(const gt_args a) {{')
    L.append("    unsigned gt_bx, gt_by, gt_bz;")
    L.append(f"    gt_tile({int(TUNING['shared_xcd_rows'])}u, gt_bx, gt_by, gt_bz);")
    L.append("    const int lane = threadIdx.x & 63;  // waves lie along I and overlap by the halo lanes")
    L.append(f"    const gt_i64 wave_x = (gt_i64)gt_bx * {block[0] // 64} + (threadIdx.x >> 6);")
    # a.lead: the origins lie `lead` items past a 16-byte boundary (see _emit_vector_kernel): the lanes start that far west
    L.append(f"    const gt_i64 i0 = (wave_x * {out_lanes} - {halo} + lane) * {vec} - a.lead;")
    L.append("    const gt_i64 iend = a.dI, jend = a.dJ;")
    L.append(f"    const gt_i64 j0 = ((gt_i64)gt_by * {block[1]} + threadIdx.y) * {JT};")
    L.append(f"    if (wave_x * {out_lanes * vec} - a.lead >= iend || j0 >= jend) return;  // whole waves only: DPP needs every lane")
    L.append(f"    const bool out_lane = lane >= {halo} && lane < {64 - halo};")
    for n in globals_:
        c = _c_ident(n)
        ct = _CTYPE[em.decl_dtype[n].name]
        const = "" if n in stage.written else "const "
        qual = " __restrict__" if n in plan.scratch else " GT_RESTRICT"
        L.append(f"    {const}{ct}* const{qual} b_{c} = a.{c} + i0 + j0 * a.{c}_sj;")
    L.append(f"    const bool whole = j0 + {JT} <= jend;  // else: a partial strip, point by point")
    L.append("    const gt_i64 k = a.k_lo + gt_bz;")
    for nest, order, defs, need in forms:
        _emit_shared_nest(em, si, stage, nest, order, defs, need, vec, JT, globals_)
    L.append("}")
    L.append("")
    return halo, tuple(globals_)


def _read_once_fields(em: "_Emitter", stage: Stage) -> Set[str]:
    """Strip kernels (TUNING["strip_nt_loads"]): the arrays a horizontal stage reads at its own point only -- every element exactly
    once, by one lane (horizontal diffusion's `coeff`).  Their 16-byte loads are nontemporal: hand-written kernel, same box A-B-A x 3,
    fp32 0.707 -> 0.728 of the HBM peak, fp64 0.707 -> 0.713; the same hint on `in`, whose halo rows other strips re-read, costs 18 %
    (profiles/r5_nt_loads_column_kernels.txt)."""
    if not int(TUNING["strip_nt_loads"]):
        return set()
    offsets: Dict[str, Set[Tuple[int, int, int]]] = {}
    for nest in stage.nests:
        for st in nest.stmts:
            for e in _stmt_field_reads(st):
                offsets.setdefault(e.name, set()).add(tuple(e.offset) if e.koffset is None else (9, 9, 9))
    written = {st.target.name for n in stage.nests for st in n.stmts}
    return {n for n, offs in offsets.items() if offs == {(0, 0, 0)} and n not in written and n not in em.plan.locals
            and n in em.decl_dtype and em.decl_dtype[n].kind in "fiu" and em.decl_dtype[n].itemsize >= 4}


def _emit_shared_nest(em: "_Emitter", si: int, stage: Stage, nest: Nest, order, defs, need, vec: int, JT: int, globals_) -> None:
    """One interval block of a `_vecs` kernel (see _emit_shared_kernel)."""
    L = em.lines
    plan = em.plan
    stmts = [obj for kind, obj in order if kind == "stmt"]
    written = {s.target.name for n in stage.nests for s in n.stmts}
    L.append(f"    if (k >= {em.bound(nest.interval.start)} && k < {em.bound(nest.interval.end)} && k < a.k_hi) {{")
    L.append("      if (whole) {")

    rows: Dict[Tuple[str, int, int], str] = {}  # (input, row, dk) -> vector register
    values: Dict[Tuple[str, int], List[str]] = {}  # (version, row) -> registers of the lane's own components
    shifts: Dict[Tuple, str] = {}
    guards: Dict[Tuple[int, int], str] = {}
    state = {"row": 0, "comp": 0}

    def tag(n: int) -> str:
        return f"m{-n}" if n < 0 else f"p{n}"

    def input_row(name: str, row: int, dk: int) -> str:
        key = (name, row, dk)
        if key not in rows:
            c, ct = _c_ident(name), _CTYPE[em.decl_dtype[name].name]
            var = f"r{len(rows)}"
            (ilo, ihi), _ = plan.field_extents.get(name, analysis.ZERO_EXTENT)
            if (ilo, ihi) not in guards:  # which of the lane's columns lie inside what the array is accessed on: once
                g = f"g{len(guards)}"
                L.append(f"        const bool {g} = i0 >= {ilo} && i0 + {vec} <= iend + ({ihi});")
                for v in range(vec):
                    L.append(f"        const bool {g}_{v} = i0 + {v} >= {ilo} && i0 + {v} < iend + ({ihi});")
                guards[(ilo, ihi)] = g
            g = guards[(ilo, ihi)]
            kterm = f"(k{dk:+d})" if dk else "k"
            L.append(f"        const {ct}* const p_{var} = b_{c} + ({kterm} * a.{c}_sk + {row} * a.{c}_sj);")
            L.append(f"        gt_vec<{ct}, {vec}> {var};")
            if name in _read_once_fields(em, stage):
                L.append(f"        if ({g}) {var} = __builtin_nontemporal_load(reinterpret_cast<const gt_vec<{ct}, {vec}>*>(p_{var}));")
            else:
                L.append(f"        if ({g}) {var} = *reinterpret_cast<const gt_vec<{ct}, {vec}>*>(p_{var});")
            L.append("        else {")
            for v in range(vec):
                L.append(f"          {var}[{v}] = {g}_{v} ? p_{var}[{v}] : ({ct})0;")
            L.append("        }")
            rows[key] = var
        return rows[key]

    def component(kind: str, ident, ct: str, regs, comp: int) -> str:
        """Component `comp` of a row of values held per lane in `regs` (own components 0 .. vec-1)."""
        if 0 <= comp < vec:
            return regs(comp)
        key = (kind, ident, comp)
        if key not in shifts:
            var = f"s{len(shifts)}"
            if comp < 0:  # from the lane below: its component vec + comp
                L.append(f"        const {ct} {var} = gt_shift<{ct}, true, true>({regs(vec + comp)});")
            else:  # from the lane above: its component comp - vec
                L.append(f"        const {ct} {var} = gt_shift<{ct}, false, true>({regs(comp - vec)});")
            shifts[key] = var
        return shifts[key]

    def ensure(version: str, row: int) -> None:
        """Emit the lane's own components of `version` on `row` (and, through the lookups, whatever they depend on)
        the first time somebody asks: values are computed in the order the output rows need them, so a row of a
        temporary is dead as soon as the last output row that reads it is done -- not kept until every row of every
        temporary exists."""
        if (version, row) in values:
            return
        ct = _CTYPE[np.dtype(defs[version].dtype).name]
        saved = dict(state)
        regs = []
        for v in range(vec):
            state["row"], state["comp"] = row, v
            val = em.expr(defs[version], "k", si, {})
            var = f"t_{_c_ident(version)}_{tag(row)}_{v}"
            L.append(f"        const {ct} {var} = {val};")
            regs.append(var)
        values[(version, row)] = regs
        state.update(saved)

    def lookup(e: ir.FieldAccess):
        row, comp = state["row"] + e.offset[1], state["comp"] + e.offset[0]
        if e.name in defs:
            ensure(e.name, row)
            regs = values[(e.name, row)]
            ct = _CTYPE[np.dtype(defs[e.name].dtype).name]
            return component("t", (e.name, row), ct, lambda c: regs[c], comp)
        if e.name in written or e.name not in em.global_names:
            return None
        var = input_row(e.name, row, e.offset[2])
        ct = _CTYPE[em.decl_dtype[e.name].name]
        return component("r", var, ct, lambda c: f"{var}[{c}]", comp)

    em.shared_lookup = lookup
    try:
        for row in range(JT):
            for obj in stmts:
                tname = obj.target.name
                c, ct = _c_ident(tname), _CTYPE[em.decl_dtype[tname].name]
                vals, conds = [], []
                for v in range(vec):
                    state["row"], state["comp"] = row, v
                    vals.append(em.expr(obj.value, "k", si, {}))
                    if obj.mask is not None:
                        conds.append(em.expr(obj.mask, "k", si, {}))
                sid = stmts.index(obj)
                names = []
                for v, val in enumerate(vals):
                    L.append(f"        const {ct} w{sid}_{c}_{row}_{v} = {val};")
                    names.append(f"w{sid}_{c}_{row}_{v}")
                ptr = f"b_{c} + (k * a.{c}_sk + {row} * a.{c}_sj)"
                if obj.mask is not None:  # under a run-time `if`: point by point where the condition holds
                    L.append("        if (out_lane) {")
                    for v in range(vec):
                        L.append(f"          if (i0 + {v} >= 0 && i0 + {v} < iend && ({conds[v]})) ({ptr})[{v}] = {names[v]};")
                    L.append("        }")
                    continue
                store = (f"__builtin_nontemporal_store(gt_vec<{ct}, {vec}>{{{', '.join(names)}}}, reinterpret_cast<gt_vec<{ct}, {vec}>*>({ptr}))"
                         if tname in em.streaming else
                         f"*reinterpret_cast<gt_vec<{ct}, {vec}>*>({ptr}) = gt_vec<{ct}, {vec}>{{{', '.join(names)}}}")
                L.append("        if (out_lane) {")
                L.append(f"          if (i0 >= 0 && i0 + {vec} <= iend) {store};")
                L.append("          else {")
                for v in range(vec):
                    L.append(f"            if (i0 + {v} >= 0 && i0 + {v} < iend) ({ptr})[{v}] = {names[v]};")
                L.append("          }")
                L.append("        }")
    finally:
        em.shared_lookup = None
    L.append("      } else if (out_lane) {")
    L.append(f"        for (int jv = 0; jv < {JT}; ++jv) {{")
    L.append("          const gt_i64 j = j0 + jv;")
    L.append("          if (j >= jend) break;")
    L.append(f"          for (int v = 0; v < {vec}; ++v) {{")
    L.append("            const gt_i64 i = i0 + v;")
    L.append("            if (i < 0) continue;")
    L.append("            if (i >= iend) break;")
    for n in globals_:
        c = _c_ident(n)
        ct = _CTYPE[em.decl_dtype[n].name]
        const = "" if n in stage.written else "const "
        L.append(f"            {const}{ct}* const t_{c} = b_{c} + v + jv * a.{c}_sj;")
    em.base_prefix = "t_"
    em.local_decls(nest, "            ")
    em.statements(nest.stmts, stage, si, "k", {}, "            ")
    em.base_prefix = "b_"
    L.append("          }")
    L.append("        }")
    L.append("      }")
    L.append("    }")


def _emit_vector_kernel(em: "_Emitter", si: int, stage: Stage, kname: str, vec: int, rows_per_lane: int, block,
                        k_per_thread: int, xcd_rows: int = 0) -> Tuple[str, ...]:
    """``<kname>_vec``: a lane owns ``vec`` consecutive I points times ``rows_per_lane`` consecutive J rows.
    Returns the arrays it touches."""
    L = em.lines
    plan = em.plan
    JT = rows_per_lane
    (_, ihi), (jlo, jhi) = stage.extent
    globals_ = [n for n in em.stage_globals(stage) if n not in plan.register_only]

    def has(n: str, axis: str) -> bool:
        return axis in em.axes.get(n, ("I", "J", "K"))

    # the workgroup shape is the LAUNCH's (blockDim): the host may add a fifth wave along I where one extra lane -- a domain
    # of 512 columns from an odd origin needs 257 -- would otherwise cost a second, almost empty workgroup per row
    L.append(f'extern "C" __global__ void __launch_bounds__({max(block[0] * block[1], 320)}) {kname}_vec(const gt_args a) {{')
    L.append("    unsigned gt_bx, gt_by, gt_bz;")
    L.append(f"    gt_tile({xcd_rows}u, gt_bx, gt_by, gt_bz);")
    L.append("    const int lane = threadIdx.x & 63;  // waves lie along I: blockDim.x is a multiple of 64")
    # a.lead: the origins lie `lead` items past a 16-byte boundary (all arrays alike; the host checks): lanes start that far
    # BEFORE the domain, so that every lane's vector is naturally aligned; the first lane of a row then holds a partial
    # vector (handled point by point, like the last one) and its right-hand neighbour loads its west column itself
    L.append(f"    const gt_i64 i0 = ((gt_i64)gt_bx * blockDim.x + threadIdx.x) * {vec} - a.lead;")
    L.append(f"    const gt_i64 iend = a.dI + ({ihi}), jend = a.dJ + ({jhi});")
    L.append(f"    const gt_i64 j0 = ((gt_i64)gt_by * blockDim.y + threadIdx.y) * {JT} + ({jlo});")
    L.append("    if (i0 >= iend || j0 >= jend) return;")
    for n in globals_:
        c = _c_ident(n)
        ct = _CTYPE[em.decl_dtype[n].name]
        const = "" if n in stage.written else "const "
        qual = " __restrict__" if n in plan.scratch else " GT_RESTRICT"
        off = " + ".join(t for t in ("i0" if has(n, "I") else "", f"j0 * a.{c}_sj" if has(n, "J") else "") if t) or "0"
        L.append(f"    {const}{ct}* const{qual} b_{c} = a.{c} + {off};")
    L.append(f"    const bool whole = i0 >= 0 && i0 + {vec} <= iend && j0 + {JT} <= jend;  // else: a partial vector / strip")
    L.append(f"    const bool edge_lo = lane == 0 || i0 < {vec}, edge_hi = lane == 63 || i0 + {2 * vec} > iend;")
    if k_per_thread > 1:
        L.append("    #pragma unroll")
        L.append(f"    for (int kk = 0; kk < {k_per_thread}; ++kk) {{")
        L.append(f"    const gt_i64 k = a.k_lo + (gt_i64)gt_bz * {k_per_thread} + kk;")
    else:
        L.append("    const gt_i64 k = a.k_lo + gt_bz;")
    for nest in stage.nests:
        L.append(f"    if (k >= {em.bound(nest.interval.start)} && k < {em.bound(nest.interval.end)} && k < a.k_hi) {{")
        L.append("      if (whole) {")
        # rows: every (array, row offset, dk) read for the strip, with the element offsets needed around the lane
        rows: Dict[Tuple[str, int, int], Set[int]] = {}
        for s in nest.stmts:
            for e in _stmt_field_reads(s):
                if e.name in plan.locals or not has(e.name, "I"):
                    continue
                for jv in range(JT if has(e.name, "J") else 1):
                    need = rows.setdefault((e.name, e.offset[1] + jv, e.offset[2]), set())
                    for v in range(vec):
                        need.add(v + e.offset[0])
        names: Dict[Tuple[str, int, int], Dict[int, str]] = {}
        # Every row's 16-byte load is issued BEFORE the first lane shift: a shift waits for its row, and the edge lanes'
        # scalar loads sit in branches the scheduler does not move loads across -- interleaved row by row (as this was
        # emitted until round 3) the later rows' loads queued up behind them (the hand-written lap5_strip_tile issues all
        # LJ + 2 rows up front: +8 % there, profiles/r1_microbench_*.log).
        after_loads: List[str] = []
        for rn, ((name, dj, dk), need) in enumerate(sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][2], kv[0][1]))):
            c = _c_ident(name)
            ct = _CTYPE[em.decl_dtype[name].name]
            base = em.access(ir.FieldAccess(name, (0, dj, dk)), "k", -1, {})  # b_x[...] of element 0
            addr = base[base.index("[") + 1:-1]
            L.append(f"        const {ct}* const p{rn} = b_{c} + ({addr});")
            if name in _read_once_fields(em, stage):
                L.append(f"        const gt_vec<{ct}, {vec}> r{rn} = __builtin_nontemporal_load(reinterpret_cast<const gt_vec<{ct}, {vec}>*>(p{rn}));")
            else:
                L.append(f"        const gt_vec<{ct}, {vec}> r{rn} = *reinterpret_cast<const gt_vec<{ct}, {vec}>*>(p{rn});")
            elems = {v: f"r{rn}[{v}]" for v in range(vec)}
            for e in sorted(x for x in need if x < 0):  # from the lane below: its component vec + e
                var = f"r{rn}_m{-e}"
                after_loads.append(f"        {ct} {var} = gt_shift<{ct}, true>(r{rn}[{vec + e}]);")
                after_loads.append(f"        if (edge_lo) {var} = p{rn}[{e}];")
                elems[e] = var
            for e in sorted(x for x in need if x >= vec):  # from the lane above: its component e - vec
                var = f"r{rn}_p{e}"
                after_loads.append(f"        {ct} {var} = gt_shift<{ct}, false>(r{rn}[{e - vec}]);")
                after_loads.append(f"        if (edge_hi) {var} = p{rn}[{e}];")
                elems[e] = var
            names[(name, dj, dk)] = elems
        L.extend(after_loads)
        em.vec_rows = names
        for jv in range(JT):
            for v in range(vec):
                em.local_suffix = f"_{jv}_{v}"
                em.local_decls(nest, "        ")

        def target_of(tname: str, jv: int) -> str:
            saved, em.vec_rows = em.vec_rows, None  # the target is addressed in memory
            t = em.access(ir.FieldAccess(tname, (0, jv if has(tname, "J") else 0, 0)), "k", -1, {})
            em.vec_rows = saved
            return t

        for sn, s in enumerate(nest.stmts):
            tname = s.target.name
            uniform = s.mask is None and s.region is None and em.guard(s, stage) is None
            for jv in range(JT):
                values = []
                for v in range(vec):
                    em.vec_component, em.vec_row, em.local_suffix = v, jv, f"_{jv}_{v}"
                    ivar, jvar = f"(i0 + {v})", f"(j0 + {jv})"
                    if tname in plan.locals:
                        g = em.full_guard(s, stage, si, "k", {}, ivar, jvar)
                        val = em.expr(s.value, "k", si, {})
                        L.append(f"        {'if (' + g + ') ' if g else ''}l_{_c_ident(tname)}_{jv}_{v} = {val};")
                        continue
                    ct = _CTYPE[em.decl_dtype[tname].name]
                    val = em.expr(s.value, "k", si, {})
                    if uniform:
                        L.append(f"        const {ct} w{sn}_{jv}_{v} = {val};")
                        values.append(f"w{sn}_{jv}_{v}")
                    else:
                        g = em.full_guard(s, stage, si, "k", {}, ivar, jvar)
                        target = target_of(tname, jv)[:-1] + f" + {v}]"
                        L.append(f"        {'if (' + g + ') ' if g else ''}{target} = {val};")
                if values:
                    ct = _CTYPE[em.decl_dtype[tname].name]
                    packed = f"gt_vec<{ct}, {vec}>{{{', '.join(values)}}}"
                    ptr = f"reinterpret_cast<gt_vec<{ct}, {vec}>*>(&{target_of(tname, jv)})"
                    if tname in em.streaming:
                        L.append(f"        __builtin_nontemporal_store({packed}, {ptr});")
                    else:
                        L.append(f"        *{ptr} = {packed};")
        em.vec_rows, em.local_suffix, em.vec_component, em.vec_row = None, "", 0, 0
        L.append("      } else {")
        L.append(f"        for (int jv = 0; jv < {JT}; ++jv) {{")
        L.append("          const gt_i64 j = j0 + jv;")
        L.append("          if (j >= jend) break;")
        L.append(f"          for (int v = 0; v < {vec}; ++v) {{")
        L.append("            const gt_i64 i = i0 + v;")
        L.append("            if (i < 0) continue;")
        L.append("            if (i >= iend) break;")
        for n in globals_:
            c = _c_ident(n)
            ct = _CTYPE[em.decl_dtype[n].name]
            const = "" if n in stage.written else "const "
            step = " + ".join(t for t in ("v" if has(n, "I") else "", f"jv * a.{c}_sj" if has(n, "J") else "") if t) or "0"
            L.append(f"            {const}{ct}* const t_{c} = b_{c} + {step};")
        em.base_prefix = "t_"
        em.local_decls(nest, "            ")
        em.statements(nest.stmts, stage, si, "k", {}, "            ")
        em.base_prefix = "b_"
        L.append("          }")
        L.append("        }")
        L.append("      }")
        L.append("    }")
    if k_per_thread > 1:
        L.append("    }")
    L.append("}")
    L.append("")
    return tuple(globals_)


def generate(stencil: ir.Stencil) -> GeneratedProgram:
    """Typed IR -> HIP source + the ctypes mirror of its argument struct."""
    plan = plan_stages(stencil)
    em = _Emitter(plan)
    fields_c: List[Tuple[str, object]] = []
    struct_lines = ["struct gt_args {"]
    for n in em.global_names:
        c = _c_ident(n)
        ct = _CTYPE[em.decl_dtype[n].name]
        struct_lines.append(f"    {ct}* {c}; gt_i64 {c}_si, {c}_sj, {c}_sk;")
        fields_c += [(c, ctypes.c_void_p), (f"{c}_si", ctypes.c_int64), (f"{c}_sj", ctypes.c_int64),
                     (f"{c}_sk", ctypes.c_int64)]
        for dn in range(len(em.data_dims.get(n, ()))):  # element strides of the data dimensions
            struct_lines.append(f"    gt_i64 {c}_d{dn};")
            fields_c.append((f"{c}_d{dn}", ctypes.c_int64))
    for p in plan.params:
        dt = np.dtype(p.dtype)
        struct_lines.append(f"    {_CTYPE[dt.name]} p_{_c_ident(p.name)};")
        fields_c.append((f"p_{_c_ident(p.name)}", _CTYPES_TYPE[dt.name]))
    struct_lines.append("    gt_i64 dI, dJ, dK;")
    struct_lines.append("    gt_i64 k_lo, k_hi;  // thread-per-point kernels cover the levels [k_lo, k_hi) of this launch")
    struct_lines.append("    gt_i64 lead;  // `_vec` kernels: the arrays' origins lie this many items past a 16-byte boundary (0 .. vec - 1)")
    struct_lines.append("};")
    fields_c += [("dI", ctypes.c_int64), ("dJ", ctypes.c_int64), ("dK", ctypes.c_int64), ("k_lo", ctypes.c_int64),
                 ("k_hi", ctypes.c_int64), ("lead", ctypes.c_int64)]
    args_struct = type("gt_args", (ctypes.Structure,), {"_fields_": fields_c})

    kernels = []
    for si, stage in enumerate(plan.stages):
        kernels.append(em.kernel(si, stage, f"gt4mi_{_c_ident(stencil.name)}_stage{si}"))
    header = [
        PRELUDE,
        "#ifdef GT4MI_UNIT_I_STRIDE",
        "#define GT_SI(x) 1",
        "#else",
        "#define GT_SI(x) (x)",
        "#endif",
        "#ifdef GT4MI_NO_ALIAS",
        "#define GT_RESTRICT __restrict__",
        "#else",
        "#define GT_RESTRICT",
        "#endif",
        "",
    ]
    source = "\n".join(header + struct_lines + [""] + em.lines)
    return GeneratedProgram(source, kernels, args_struct, plan, set(em.inexact))
