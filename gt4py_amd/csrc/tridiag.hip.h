// Vertical tridiagonal (Thomas) solve for gfx950: one thread per (i,j) column, K serial per thread.
//
// Reference semantics: `tridiagonal_solver`
//   /root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/stencil_definitions.py:219-232
// as executed by the numpy backend: the FORWARD sweep rewrites sup/rhs
// in place level by level, the BACKWARD sweep fills out.  All statements have zero horizontal
// offsets, so plane-by-plane statement order is identical to independent per-column recurrences.
//
//   k = 0      : sup = sup/diag ; rhs = rhs/diag
//   k >= 1     : den = diag - sup[k-1]*inf         (updated sup[k-1]; evaluated twice, same value)
//                sup = sup/den ; rhs = (rhs - inf*rhs[k-1])/den
//   k = K-1    : out = rhs
//   k < K-1    : out = rhs - sup*out[k+1]
//
// IEEE-correct division (hipcc's default f32/f64 divide expansion is correctly rounded; the build
// must not use -ffast-math) and no FMA contraction keep this bit-identical to numpy.
//
// Roofline: HBM.  Algorithmic traffic 7*sizeof(T) bytes per lattice update (read inf, diag, sup,
// rhs; write sup, rhs, out).  This two-sweep kernel re-reads sup', rhs' in the backward sweep
// (9*sizeof(T) moved per update, minus whatever the top levels still find in L2).
//
// Lanes run along I (contiguous), VEC columns per lane, so every level is a coalesced row segment;
// loads for the next UNROLL levels are issued before the dependent arithmetic of the current ones.
#pragma once

#include "common.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

template <typename T, int VEC, int UNROLL>
__global__ void __launch_bounds__(256)
tridiag_kernel(View<const T> inf, View<const T> diag, View<T> sup, View<T> rhs, View<T> out, int dI,
               int dJ, int dK, unsigned tiles_i) {
    const unsigned bi = blockIdx.x % tiles_i;
    const unsigned j = blockIdx.x / tiles_i;
    const int i0 = (int)(bi * 256 + threadIdx.x) * VEC;
    if (i0 >= dI) return;

    const T* __restrict__ p_inf = inf.p + (int64_t)j * inf.sj + i0;
    const T* __restrict__ p_diag = diag.p + (int64_t)j * diag.sj + i0;
    T* __restrict__ p_sup = sup.p + (int64_t)j * sup.sj + i0;
    T* __restrict__ p_rhs = rhs.p + (int64_t)j * rhs.sj + i0;
    T* __restrict__ p_out = out.p + (int64_t)j * out.sj + i0;

    T sp[VEC], rp[VEC];  // updated sup[k-1], rhs[k-1]

    // ---- FORWARD ------------------------------------------------------------------------------
    {
        T d[VEC], s[VEC], r[VEC];
        vload<T, VEC>(p_diag, d);
        vload<T, VEC>(p_sup, s);
        vload<T, VEC>(p_rhs, r);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sp[e] = s[e] / d[e];
            rp[e] = r[e] / d[e];
        }
        vstore<T, VEC, false>(p_sup, sp);
        vstore<T, VEC, false>(p_rhs, rp);
    }
    auto fwd_level = [&](int k, const T (&a)[VEC], const T (&d)[VEC], const T (&s)[VEC],
                         const T (&r)[VEC]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const T den1 = d[e] - (sp[e] * a[e]);
            const T ns = s[e] / den1;
            const T num = r[e] - (a[e] * rp[e]);
            const T den2 = d[e] - (sp[e] * a[e]);
            const T nr = num / den2;
            sp[e] = ns;
            rp[e] = nr;
        }
        vstore<T, VEC, false>(p_sup + (int64_t)k * sup.sk, sp);
        vstore<T, VEC, false>(p_rhs + (int64_t)k * rhs.sk, rp);
    };
    int k = 1;
    for (; k + UNROLL <= dK; k += UNROLL) {
        T a[UNROLL][VEC], d[UNROLL][VEC], s[UNROLL][VEC], r[UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            vload<T, VEC>(p_inf + (int64_t)(k + u) * inf.sk, a[u]);
            vload<T, VEC>(p_diag + (int64_t)(k + u) * diag.sk, d[u]);
            vload<T, VEC>(p_sup + (int64_t)(k + u) * sup.sk, s[u]);
            vload<T, VEC>(p_rhs + (int64_t)(k + u) * rhs.sk, r[u]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) fwd_level(k + u, a[u], d[u], s[u], r[u]);
    }
    for (; k < dK; ++k) {
        T a[VEC], d[VEC], s[VEC], r[VEC];
        vload<T, VEC>(p_inf + (int64_t)k * inf.sk, a);
        vload<T, VEC>(p_diag + (int64_t)k * diag.sk, d);
        vload<T, VEC>(p_sup + (int64_t)k * sup.sk, s);
        vload<T, VEC>(p_rhs + (int64_t)k * rhs.sk, r);
        fwd_level(k, a, d, s, r);
    }

    // ---- BACKWARD -----------------------------------------------------------------------------
    T o[VEC];
    int kb = dK - 1;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = rp[e];
    vstore<T, VEC, true>(p_out + (int64_t)kb * out.sk, o);
    kb = dK - 2;
    for (; kb - UNROLL + 1 >= 0; kb -= UNROLL) {
        T s[UNROLL][VEC], r[UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            vload<T, VEC>(p_sup + (int64_t)(kb - u) * sup.sk, s[u]);
            vload<T, VEC>(p_rhs + (int64_t)(kb - u) * rhs.sk, r[u]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = r[u][e] - (s[u][e] * o[e]);
            vstore<T, VEC, true>(p_out + (int64_t)(kb - u) * out.sk, o);
        }
    }
    for (; kb >= 0; --kb) {
        T s[VEC], r[VEC];
        vload<T, VEC>(p_sup + (int64_t)kb * sup.sk, s);
        vload<T, VEC>(p_rhs + (int64_t)kb * rhs.sk, r);
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = r[e] - (s[e] * o[e]);
        vstore<T, VEC, true>(p_out + (int64_t)kb * out.sk, o);
    }
}

// Any-stride fallback: one thread per column, scalar accesses.
template <typename T>
__global__ void __launch_bounds__(256)
tridiag_generic_kernel(View<const T> inf, View<const T> diag, View<T> sup, View<T> rhs, View<T> out,
                       int dI, int dJ, int dK) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= dI || j >= dJ) return;
    const T* pa = inf.p + (int64_t)i * inf.si + (int64_t)j * inf.sj;
    const T* pd = diag.p + (int64_t)i * diag.si + (int64_t)j * diag.sj;
    T* ps = sup.p + (int64_t)i * sup.si + (int64_t)j * sup.sj;
    T* pr = rhs.p + (int64_t)i * rhs.si + (int64_t)j * rhs.sj;
    T* po = out.p + (int64_t)i * out.si + (int64_t)j * out.sj;
    T sp = ps[0] / pd[0];
    T rp = pr[0] / pd[0];
    ps[0] = sp;
    pr[0] = rp;
    for (int k = 1; k < dK; ++k) {
        const T a = pa[(int64_t)k * inf.sk], d = pd[(int64_t)k * diag.sk];
        const T den1 = d - (sp * a);
        const T ns = ps[(int64_t)k * sup.sk] / den1;
        const T num = pr[(int64_t)k * rhs.sk] - (a * rp);
        const T den2 = d - (sp * a);
        rp = num / den2;
        sp = ns;
        ps[(int64_t)k * sup.sk] = sp;
        pr[(int64_t)k * rhs.sk] = rp;
    }
    T o = rp;
    po[(int64_t)(dK - 1) * out.sk] = o;
    for (int k = dK - 2; k >= 0; --k) {
        o = pr[(int64_t)k * rhs.sk] - (ps[(int64_t)k * sup.sk] * o);
        po[(int64_t)k * out.sk] = o;
    }
}

}  // namespace gt4mi

#include "tridiag_stack.hip.h"

namespace gt4mi {

struct TridiagTuning {
    // profiles/r1_microbench_d_*.log: 8 bytes per lane with 8 levels of loads in flight beats
    // 16-byte lanes (84 vs 78 GLUPS on 1024x1024x160 f64).
    static constexpr int UNROLL = 8;
    // on-chip stack of the forward sweep's results (tridiag_stack.hip.h): the last 32 levels in registers,
    // the 40 before them in LDS; 16/24/48 register levels and 0..64 LDS levels measured within 2 % of each
    // other, more than 48 register levels slower (profiles/r1_microbench_i_tridiag_stack.log).  The launched
    // kernel is the software-pipelined form (tridiag_pipe_kernel, +2 % on the same box and bit-identical:
    // profiles/r2_tridiag_pipelined.log); for it 40 LDS levels are also what keeps ONE wave per SIMD, fewer were
    // 15-25 % slower.
    static constexpr int STACK_REG = 32, STACK_LDS = 40, STACK_U = 8;
    // round 2: with a scheduling fence per unrolled batch more register levels pay again -- 80 + 40 levels on chip
    // (478 registers, no scratch): 84.5 GLUPS next to 81.4 for 32 + 40 on the same box, 60 instead of 64.8 B/LUP moved
    // (profiles/r2_tridiag_pipelined_deep.log).  Used when the column is deeper than 120 levels.
    static constexpr int STACK_REG_DEEP = 80;
    // ... and 104 + 40 in batches of 4 (502 registers, no scratch) for columns deeper than 144 levels: 95.8 GLUPS next to
    // 94.0 for 80 + 40 on the same box, 1.03x instead of 1.07x the algorithmic traffic (profiles/r2_microbench_tripipe_deep.log).
    // Needs the larger -pragma-unroll-threshold of the Makefile: with LLVM's default the 36-batch level loop stays
    // rolled and the register arrays become scratch (tests/test_c_abi.py reads the compiler's resource remarks).
    static constexpr int STACK_REG_DEEPER = 104, STACK_U_DEEPER = 4;
    // round 5, with the nontemporal loads: TWO waves (two J rows) per workgroup -- 1.582-1.592 ms against 1.624-1.634 on one box, 1.713-1.721
    // against 1.748-1.755 on another (+2-2.6 %, A-B x 3 each; four waves +1 %; without the nontemporal loads the same change is +0.6 %;
    // profiles/r5_nt_loads_column_kernels.txt).  80 KB of LDS per workgroup, still one wave per SIMD.
    static constexpr int STACK_WPB_DEEPER = 2;
    // ... and in the shallower variants as well: K = 80 (32 + 40 levels) 0.878 -> 0.867 ms, K = 60 (16 + 40) 0.628 -> 0.615 ms (microbench trint,
    // MB_TRINT_SHALLOW=1, A-B x 3; the nontemporal loads themselves: 0.929 -> 0.878 and 0.682 -> 0.628)
    static constexpr int STACK_WPB = 2;
    static constexpr int STACK_REG_SHALLOW = 16;  // + 40 LDS levels, for columns of 57 ... 72 levels
};

template <typename T>
inline int tridiag_run(const int64_t domain[3], const gt4mi_field* inf, const gt4mi_field* diag,
                       const gt4mi_field* sup, const gt4mi_field* rhs, const gt4mi_field* out,
                       hipStream_t stream) {
    if (int rc = check_domain(domain)) return rc;
    if (domain[2] < 2)
        return fail(GT4MI_ERR_INVALID_ARGUMENT,
                    "tridiag: sequential axis is %lld, but must be at least 2", (long long)domain[2]);
    const int h0[3] = {0, 0, 0};
    View<T> a, d, s, r, o;
    if (int rc = make_view<T>("inf", inf, domain, h0, h0, &a)) return rc;
    if (int rc = make_view<T>("diag", diag, domain, h0, h0, &d)) return rc;
    if (int rc = make_view<T>("sup", sup, domain, h0, h0, &s)) return rc;
    if (int rc = make_view<T>("rhs", rhs, domain, h0, h0, &r)) return rc;
    if (int rc = make_view<T>("out", out, domain, h0, h0, &o)) return rc;
    if (domain[0] == 0 || domain[1] == 0) return GT4MI_OK;
    // Aliases.  Every access is at zero horizontal offset and a thread owns its column, so what matters is the
    // order of accesses within a column.  `out` may BE one of the other fields (same elements): out[k] is written
    // after everything that level still needs of the aliased field has been read, level by level as the
    // reference does -- by the column-at-a-time kernel without __restrict__ and without on-chip copies.  Any
    // other overlap with a written field (sup with rhs, sup or rhs with inf / diag, shifted views) changes what
    // the reference's statement-by-statement evaluation reads and is refused.
    bool alias = false;
    {
        const View<T>* views[5] = {&a, &d, &s, &r, &o};
        const char* names[5] = {"inf", "diag", "sup", "rhs", "out"};
        for (int w = 2; w < 5; ++w)          // written fields: sup, rhs, out
            for (int x = 0; x < 5; ++x) {
                if (x == w || (x > w && x >= 2)) continue;  // written/written pairs once
                // (element-disjoint views of one buffer -- interleaved slices, halves -- are not an overlap)
                if (!views_overlap(*views[w], h0, h0, *views[x], h0, h0, domain)) continue;
                if (w == 4 && same_view(*views[w], *views[x])) {
                    alias = true;
                    continue;
                }
                return fail(GT4MI_ERR_UNSUPPORTED, "tridiag: '%s' and '%s' overlap in memory; only 'out' may share its array "
                                                   "(element for element) with another field", names[w], names[x]);
            }
    }
    const View<const T> ac{a.p, a.si, a.sj, a.sk}, dc{d.p, d.si, d.sj, d.sk};
    const bool contiguous = !alias && a.si == 1 && d.si == 1 && s.si == 1 && r.si == 1 && o.si == 1;
    if (contiguous && domain[2] > TridiagTuning::STACK_REG) {
        // keep the top of the column on chip between the sweeps
        const unsigned ti = (unsigned)cdiv(domain[0], 64);
        // (8-byte items only: for float the LDS share allows two waves per SIMD, the compiler then budgets 256 registers
        // and the deep variant spills)
        constexpr int DEEP = sizeof(T) == 8 ? TridiagTuning::STACK_REG_DEEP : TridiagTuning::STACK_REG;
        if (sizeof(T) == 8 && domain[2] > TridiagTuning::STACK_REG_DEEPER + TridiagTuning::STACK_LDS) {
            // GT4MI_TRIDIAG_NT_LOADS=0: the same kernel with plain loads; GT4MI_TRIDIAG_WPB=1: one wave per workgroup as in rounds 2-4
            // (A/B runs; nontemporal loads and two waves per workgroup are the defaults, see NTL and STACK_WPB_DEEPER)
            // (read once; with GT4MI_TRIDIAG_AB=1 at every call, so that ONE process can alternate the variants on the same fields)
            auto env_is = [](const char* name, char c) { const char* e = getenv(name); return e && e[0] == c; };
            static const bool every_call = getenv("GT4MI_TRIDIAG_AB") != nullptr;
            static const bool plain_loads_once = env_is("GT4MI_TRIDIAG_NT_LOADS", '0'), one_wave_once = env_is("GT4MI_TRIDIAG_WPB", '1');
            const bool plain_loads = every_call ? env_is("GT4MI_TRIDIAG_NT_LOADS", '0') : plain_loads_once;
            const bool one_wave = every_call ? env_is("GT4MI_TRIDIAG_WPB", '1') : one_wave_once;
            constexpr int RLD = sizeof(T) == 8 ? TridiagTuning::STACK_REG_DEEPER : TridiagTuning::STACK_REG;
            constexpr int WPB = TridiagTuning::STACK_WPB_DEEPER;
            if (plain_loads)
                hipLaunchKernelGGL((tridiag_pipe_kernel<T, RLD, TridiagTuning::STACK_LDS, TridiagTuning::STACK_U_DEEPER, 1, 0, 0>),
                                   dim3(ti * (unsigned)domain[1]), dim3(64), 0, stream, ac, dc, s, r, o, (int)domain[0],
                                   (int)domain[1], (int)domain[2], ti);
            else if (one_wave)
                hipLaunchKernelGGL((tridiag_pipe_kernel<T, RLD, TridiagTuning::STACK_LDS, TridiagTuning::STACK_U_DEEPER, 1>),
                                   dim3(ti * (unsigned)domain[1]), dim3(64), 0, stream, ac, dc, s, r, o, (int)domain[0],
                                   (int)domain[1], (int)domain[2], ti);
            else
                hipLaunchKernelGGL((tridiag_pipe_kernel<T, RLD, TridiagTuning::STACK_LDS, TridiagTuning::STACK_U_DEEPER, WPB>),
                                   dim3(ti * (unsigned)cdiv(domain[1], WPB)), dim3(64, WPB), 0, stream, ac, dc, s, r, o, (int)domain[0],
                                   (int)domain[1], (int)domain[2], ti);
        } else if (sizeof(T) == 8 && domain[2] > DEEP + TridiagTuning::STACK_LDS) {
            hipLaunchKernelGGL((tridiag_pipe_kernel<T, DEEP, TridiagTuning::STACK_LDS, TridiagTuning::STACK_U, TridiagTuning::STACK_WPB>),
                               dim3(ti * (unsigned)cdiv(domain[1], TridiagTuning::STACK_WPB)), dim3(64, TridiagTuning::STACK_WPB), 0, stream, ac, dc, s, r, o, (int)domain[0],
                               (int)domain[1], (int)domain[2], ti);
        } else if (domain[2] > TridiagTuning::STACK_REG + TridiagTuning::STACK_LDS) {
            hipLaunchKernelGGL((tridiag_pipe_kernel<T, TridiagTuning::STACK_REG, TridiagTuning::STACK_LDS, TridiagTuning::STACK_U, TridiagTuning::STACK_WPB>),
                               dim3(ti * (unsigned)cdiv(domain[1], TridiagTuning::STACK_WPB)), dim3(64, TridiagTuning::STACK_WPB), 0, stream, ac, dc, s, r, o, (int)domain[0],
                               (int)domain[1], (int)domain[2], ti);
        } else if (sizeof(T) == 8 && domain[2] > TridiagTuning::STACK_REG_SHALLOW + TridiagTuning::STACK_LDS) {
            // 57 ... 72 levels (K = 60 is a common column depth): 16 + 40 on chip instead of 32 in registers only
            hipLaunchKernelGGL((tridiag_pipe_kernel<T, TridiagTuning::STACK_REG_SHALLOW, TridiagTuning::STACK_LDS, TridiagTuning::STACK_U, TridiagTuning::STACK_WPB>),
                               dim3(ti * (unsigned)cdiv(domain[1], TridiagTuning::STACK_WPB)), dim3(64, TridiagTuning::STACK_WPB), 0, stream, ac, dc, s, r, o, (int)domain[0],
                               (int)domain[1], (int)domain[2], ti);
        } else {
            hipLaunchKernelGGL((tridiag_pipe_kernel<T, TridiagTuning::STACK_REG, 0, TridiagTuning::STACK_U, TridiagTuning::STACK_WPB>),
                               dim3(ti * (unsigned)cdiv(domain[1], TridiagTuning::STACK_WPB)), dim3(64, TridiagTuning::STACK_WPB), 0, stream, ac, dc, s, r, o, (int)domain[0],
                               (int)domain[1], (int)domain[2], ti);
        }
    } else if (contiguous) {
        constexpr int VMAX = 8 / sizeof(T);
        const bool vec = VMAX > 1 &&vec_ok(a, VMAX) && vec_ok(d, VMAX) && vec_ok(s, VMAX) && vec_ok(r, VMAX) &&
                         vec_ok(o, VMAX) && (domain[0] % VMAX == 0);
        if (vec) {
            const unsigned ti = (unsigned)cdiv(domain[0], 256 * VMAX);
            hipLaunchKernelGGL((tridiag_kernel<T, VMAX, TridiagTuning::UNROLL>),
                               dim3(ti * (unsigned)domain[1]), dim3(256), 0, stream, ac, dc, s, r, o,
                               (int)domain[0], (int)domain[1], (int)domain[2], ti);
        } else {
            const unsigned ti = (unsigned)cdiv(domain[0], 256);
            hipLaunchKernelGGL((tridiag_kernel<T, 1, TridiagTuning::UNROLL>),
                               dim3(ti * (unsigned)domain[1]), dim3(256), 0, stream, ac, dc, s, r, o,
                               (int)domain[0], (int)domain[1], (int)domain[2], ti);
        }
    } else {
        dim3 grid((unsigned)cdiv(domain[0], 64), (unsigned)cdiv(domain[1], 4));
        hipLaunchKernelGGL((tridiag_generic_kernel<T>), grid, dim3(256), 0, stream, ac, dc, s, r, o,
                           (int)domain[0], (int)domain[1], (int)domain[2]);
    }
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

}  // namespace gt4mi
