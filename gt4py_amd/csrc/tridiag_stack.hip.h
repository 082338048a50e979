// Tridiagonal solve that keeps the forward sweep's results ON CHIP for the backward sweep.
//
// tridiag.hip.h moves 9 arrays' worth of bytes per solve: the backward sweep re-reads the sup', rhs'
// the forward sweep has just written (PMC: 1.49x the algorithmic reads,
// profiles/r1_kernel_hbm_traffic_pmc.txt) and it already runs at the streaming-copy rate, so the only way
// up is to not re-read.  A CDNA4 CU has 512 KB of vector registers and 160 KB of LDS; a 160-level fp64
// column pair is 2.5 KB.  Here a wave keeps the LAST `RL` levels of (sup', rhs') in registers (the level
// loops over them are fully unrolled, so the arrays stay in VGPRs), the `LL` levels before those in LDS,
// and only the first dK - RL - LL levels are read back from memory:
//
//     forward   k in [0, A)            compute, store                      (A = dK - LL - RL)
//               k in [A, A + LL)       compute, store, keep in LDS
//               k in [A + LL, dK)      compute, store, keep in registers   (unrolled)
//     backward  the same three ranges in reverse; memory is only touched for [0, A) and for `out`.
//
// Traffic per lattice update: 7 values + 2 * A / dK values.  One wave per workgroup (LL * 1 KB of LDS each for
// fp64), so with LL = 40 four waves share a CU and memory latency is hidden by issuing the loads of `U` levels
// ahead of the dependent divides rather than by other waves.  Measured on MI355X, fp64 1024x1024x160
// (profiles/r1_microbench_i_tridiag_stack.log): RL = 32, LL = 40 (72 of 160 levels on chip, 64.8 B instead of
// 72 B per update) is 10-12 % faster than tridiag.hip.h on the same device; RL >= 48 is SLOWER again (the
// compiler does keep 96 levels in 256 VGPRs + 180 AGPRs without scratch, but the unrolled code and the
// single wave per SIMD cost more than the saved traffic brings).
//
// Values are those of tridiag.hip.h (same expressions, same order, IEEE divides, no contraction): the
// on-chip copies are the very numbers that were stored.
#pragma once

#include "common.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

template <typename T, int RL, int LL, int U>
__global__ void __launch_bounds__(64)
tridiag_stack_kernel(View<const T> inf, View<const T> diag, View<T> sup, View<T> rhs, View<T> out, int dI, int dJ,
                     int dK, unsigned tiles_i) {
    static_assert(RL % U == 0 && LL % U == 0, "level ranges are processed in batches of U");
    __shared__ T lds[LL > 0 ? LL * 2 * 64 : 1];
    const unsigned bi = blockIdx.x % tiles_i;
    const unsigned j = blockIdx.x / tiles_i;
    const int lane = threadIdx.x;
    const int i0 = (int)(bi * 64) + lane;
    if (i0 >= dI) return;

    const T* __restrict__ p_inf = inf.p + (int64_t)j * inf.sj + i0;
    const T* __restrict__ p_diag = diag.p + (int64_t)j * diag.sj + i0;
    T* __restrict__ p_sup = sup.p + (int64_t)j * sup.sj + i0;
    T* __restrict__ p_rhs = rhs.p + (int64_t)j * rhs.sj + i0;
    T* __restrict__ p_out = out.p + (int64_t)j * out.sj + i0;

    const int A = dK - LL - RL;  // >= 1 (checked by the host): level 0 is always in the memory range
    T sp, rp;                    // updated sup[k-1], rhs[k-1]
    T S[RL], R[RL];              // the last RL levels; only ever indexed by compile-time constants

    auto level = [&](T a, T d, T s, T r) {
        const T den1 = d - (sp * a);
        const T ns = s / den1;
        const T num = r - (a * rp);
        const T den2 = d - (sp * a);
        const T nr = num / den2;
        sp = ns;
        rp = nr;
    };

    // ---- FORWARD, memory range [0, A) -----------------------------------------------------------
    {
        const T d = p_diag[0], s = p_sup[0], r = p_rhs[0];
        sp = s / d;
        rp = r / d;
        __builtin_nontemporal_store(sp, p_sup);
        __builtin_nontemporal_store(rp, p_rhs);
    }
    int k = 1;
    for (; k + U <= A; k += U) {
        T a[U], d[U], s[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = p_inf[(int64_t)(k + u) * inf.sk];
            d[u] = p_diag[(int64_t)(k + u) * diag.sk];
            s[u] = p_sup[(int64_t)(k + u) * sup.sk];
            r[u] = p_rhs[(int64_t)(k + u) * rhs.sk];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            level(a[u], d[u], s[u], r[u]);
            p_sup[(int64_t)(k + u) * sup.sk] = sp;  // read again by the backward sweep: keep cacheable
            p_rhs[(int64_t)(k + u) * rhs.sk] = rp;
        }
    }
    for (; k < A; ++k) {
        level(p_inf[(int64_t)k * inf.sk], p_diag[(int64_t)k * diag.sk], p_sup[(int64_t)k * sup.sk],
              p_rhs[(int64_t)k * rhs.sk]);
        p_sup[(int64_t)k * sup.sk] = sp;
        p_rhs[(int64_t)k * rhs.sk] = rp;
    }
    // ---- FORWARD, LDS range [A, A + LL) ---------------------------------------------------------
#pragma unroll 1
    for (int l = 0; l < LL; l += U) {
        T a[U], d[U], s[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t kk = A + l + u;
            a[u] = p_inf[kk * inf.sk];
            d[u] = p_diag[kk * diag.sk];
            s[u] = p_sup[kk * sup.sk];
            r[u] = p_rhs[kk * rhs.sk];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t kk = A + l + u;
            level(a[u], d[u], s[u], r[u]);
            __builtin_nontemporal_store(sp, p_sup + kk * sup.sk);
            __builtin_nontemporal_store(rp, p_rhs + kk * rhs.sk);
            lds[((l + u) * 2 + 0) * 64 + lane] = sp;
            lds[((l + u) * 2 + 1) * 64 + lane] = rp;
        }
    }
    // ---- FORWARD, register range [A + LL, dK): fully unrolled -------------------------------------
#pragma unroll
    for (int l = 0; l < RL; l += U) {
        T a[U], d[U], s[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t kk = A + LL + l + u;
            a[u] = p_inf[kk * inf.sk];
            d[u] = p_diag[kk * diag.sk];
            s[u] = p_sup[kk * sup.sk];
            r[u] = p_rhs[kk * rhs.sk];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t kk = A + LL + l + u;
            level(a[u], d[u], s[u], r[u]);
            __builtin_nontemporal_store(sp, p_sup + kk * sup.sk);
            __builtin_nontemporal_store(rp, p_rhs + kk * rhs.sk);
            S[l + u] = sp;
            R[l + u] = rp;
        }
    }

    // ---- BACKWARD ---------------------------------------------------------------------------------
    T o = R[RL - 1];  // out[K-1] = rhs'[K-1]
    __builtin_nontemporal_store(o, p_out + (int64_t)(dK - 1) * out.sk);
#pragma unroll
    for (int l = RL - 2; l >= 0; --l) {
        o = R[l] - (S[l] * o);
        __builtin_nontemporal_store(o, p_out + (int64_t)(A + LL + l) * out.sk);
    }
#pragma unroll 4
    for (int l = LL - 1; l >= 0; --l) {
        const T s = lds[(l * 2 + 0) * 64 + lane], r = lds[(l * 2 + 1) * 64 + lane];
        o = r - (s * o);
        __builtin_nontemporal_store(o, p_out + (int64_t)(A + l) * out.sk);
    }
    int kb = A - 1;
    for (; kb - U + 1 >= 0; kb -= U) {
        T s[U], r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] = p_sup[(int64_t)(kb - u) * sup.sk];
            r[u] = p_rhs[(int64_t)(kb - u) * rhs.sk];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o = r[u] - (s[u] * o);
            __builtin_nontemporal_store(o, p_out + (int64_t)(kb - u) * out.sk);
        }
    }
    for (; kb >= 0; --kb) {
        o = p_rhs[(int64_t)kb * rhs.sk] - (p_sup[(int64_t)kb * sup.sk] * o);
        __builtin_nontemporal_store(o, p_out + (int64_t)kb * out.sk);
    }
}

// ---- software-pipelined form -----------------------------------------------------------------------
// tridiag_stack_kernel issues the loads of a batch of U levels, waits, computes (two IEEE divides per level:
// ~30 instructions x U), stores, and only then issues the next batch: with one wave per SIMD (the on-chip stack
// leaves room for no more) nothing is in flight while it computes, and the solve ran at 79-87 GLUPS on some
// boxes and 94-97 on others whatever the placement of the fields (profiles/r2_tridiag_placement_study.log).
// Here the loads of batch n + 1 are in flight while batch n is computed, in both sweeps, across the boundaries
// between the memory / LDS / register ranges, and the first loads of the backward sweep are issued before its
// on-chip part.  Same arithmetic, same order: level 0 goes through the general formula with inf := 0 and
// sup'[-1] = rhs'[-1] := 0, which is exact (d - 0*0 = d, r - 0*0 = r for every d, r including -0, inf, nan).
template <typename T, int U>
struct TridiagFwdBatch {
    T a[U], d[U], s[U], r[U];
};
template <typename T, int U>
struct TridiagBwdBatch {
    T s[U], r[U];
};

// WPB waves per workgroup, one J row each (threadIdx.y): the waves that share a CU then work on adjacent rows of the
// same I tile -- the same pages at every level -- instead of whatever tiles the dispatcher hands the CU.
// MAP: which (I tile, J row) a workgroup takes.  Workgroup b runs on XCD b % 8 (round-robin dispatch); with one wave per SIMD
// 1 024 workgroups are resident: 64 rows of a 1 024-column domain, 512 KB of every K level of every field.
//   0  b -> tile b % tiles_i of row b / tiles_i: every XCD works on two 512-byte pieces of EVERY resident row
//   1  whole rows per XCD: XCD x takes row 8 n + x (its L2 and UTCL2 see contiguous 8 KB rows)
//   2  a contiguous eighth of the rows per XCD (rows [x dJ / 8, (x + 1) dJ / 8): every XCD inside its own pages)
//   3  row-major inside bands of 256 rows = ONE 2 MiB page of every level: band by band, tile-column by tile-column
// (round 4, VERDICT item 5: profiles/r4_tridiag_translation.txt has time and translation counters of all four)
// NTL (round 5): 1 = nontemporal loads of all four input streams of the forward sweep (a column kernel reads every element exactly
// once, level after level, a plane apart: nothing is worth keeping in the L1), 2 = of the two read-only ones (inf, diag) only, 0 = plain
// loads.  1024 x 1024 x 160 fp64, same box A-B-A x 3 (experiments/microbench.hip `trint`): 1.795 ms plain, 1.738 ms with 2, 1.650 ms
// with 1 (+8.7 %), bit-identical (profiles/r5_nt_loads_column_kernels.txt).  The Laplacian LOST 6-9 % with nontemporal loads in
// round 1 (its halo rows are re-read through the L1).
template <typename T, int RL, int LL, int U, int WPB = 1, int MAP = 0, int NTL = 1>
__global__ void __launch_bounds__(64 * WPB)
tridiag_pipe_kernel(View<const T> inf, View<const T> diag, View<T> sup, View<T> rhs, View<T> out, int dI, int dJ,
                    int dK, unsigned tiles_i) {
    static_assert(RL % U == 0 && LL % U == 0 && RL >= U, "level ranges are processed in batches of U");
    __shared__ T lds_all[LL > 0 ? LL * 2 * 64 * WPB : 1];
    T* const lds = lds_all + (LL > 0 ? threadIdx.y * (LL * 2 * 64) : 0);
    unsigned b = blockIdx.x;
    if constexpr (MAP == 1) {
        const unsigned R = 8 * tiles_i, full = (gridDim.x / R) * R;
        if (b < full) b = (b / R) * R + (b % 8) * tiles_i + (b % R) / 8;
    } else if constexpr (MAP == 2) {
        b = xcd_remap(b, gridDim.x);
    }
    unsigned bi = b % tiles_i, jrow = b / tiles_i;
    if constexpr (MAP == 3) {
        const unsigned band = 256 / WPB, per_band = band * tiles_i, nb = b / per_band, r = b % per_band;
        const unsigned rows_here = min(band, (unsigned)((dJ + WPB - 1) / WPB) - nb * band);
        bi = r / rows_here;
        jrow = nb * band + r % rows_here;
    }
    const unsigned j = jrow * WPB + threadIdx.y;
    if ((int)j >= dJ) return;
    const int lane = threadIdx.x;
    const int i0 = (int)(bi * 64) + lane;
    if (i0 >= dI) return;

    const T* __restrict__ p_inf = inf.p + (int64_t)j * inf.sj + i0;
    const T* __restrict__ p_diag = diag.p + (int64_t)j * diag.sj + i0;
    T* __restrict__ p_sup = sup.p + (int64_t)j * sup.sj + i0;
    T* __restrict__ p_rhs = rhs.p + (int64_t)j * rhs.sj + i0;
    T* __restrict__ p_out = out.p + (int64_t)j * out.sj + i0;

    const int A = dK - LL - RL;  // levels [0, A) live in memory only; A >= 1 (checked by the host)
    T sp = T(0), rp = T(0);      // sup'[k-1], rhs'[k-1]
    T S[RL], R[RL];              // the last RL levels; only ever indexed by compile-time constants

    using FB = TridiagFwdBatch<T, U>;
    using BB = TridiagBwdBatch<T, U>;
    auto level = [&](T a, T d, T s, T r) {
        const T den1 = d - (sp * a);
        const T ns = s / den1;
        const T num = r - (a * rp);
        const T den2 = d - (sp * a);
        const T nr = num / den2;
        sp = ns;
        rp = nr;
    };
    auto load = [&](FB& b, int k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (NTL >= 1) {
                b.a[u] = __builtin_nontemporal_load(p_inf + (int64_t)(k + u) * inf.sk);
                b.d[u] = __builtin_nontemporal_load(p_diag + (int64_t)(k + u) * diag.sk);
            } else {
                b.a[u] = p_inf[(int64_t)(k + u) * inf.sk];
                b.d[u] = p_diag[(int64_t)(k + u) * diag.sk];
            }
            if constexpr (NTL == 1) {
                b.s[u] = __builtin_nontemporal_load(p_sup + (int64_t)(k + u) * sup.sk);
                b.r[u] = __builtin_nontemporal_load(p_rhs + (int64_t)(k + u) * rhs.sk);
            } else {
                b.s[u] = p_sup[(int64_t)(k + u) * sup.sk];
                b.r[u] = p_rhs[(int64_t)(k + u) * rhs.sk];
            }
        }
    };
    auto forward_mem = [&](const FB& b, int k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            level(b.a[u], b.d[u], b.s[u], b.r[u]);
            p_sup[(int64_t)(k + u) * sup.sk] = sp;  // read again by the backward sweep: keep cacheable
            p_rhs[(int64_t)(k + u) * rhs.sk] = rp;
        }
    };

    // ---- FORWARD ----------------------------------------------------------------------------------
    FB B[2];
    const int head = A % U;  // levels [0, head) come first so that whole batches end exactly at A
    {
        FB h;
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < head) {  // wave-uniform
                h.a[u] = u == 0 ? T(0) : p_inf[(int64_t)u * inf.sk];
                h.d[u] = p_diag[(int64_t)u * diag.sk];
                h.s[u] = p_sup[(int64_t)u * sup.sk];
                h.r[u] = p_rhs[(int64_t)u * rhs.sk];
            }
        load(B[0], head);  // the first whole batch (of the memory range, or of the on-chip ranges when A < U)
        if (head == 0) B[0].a[0] = T(0);
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < head) {
                level(h.a[u], h.d[u], h.s[u], h.r[u]);
                p_sup[(int64_t)u * sup.sk] = sp;
                p_rhs[(int64_t)u * rhs.sk] = rp;
            }
    }
    int k = head;
    // invariant: B[0] holds (or is receiving) the batch that starts at level k
    while (k + 2 * U <= A) {
        load(B[1], k + U);
        forward_mem(B[0], k);
        load(B[0], k + 2 * U);  // may already be the first on-chip batch: same form, levels are contiguous
        forward_mem(B[1], k + U);
        k += 2 * U;
    }
    if (k + U <= A) {  // an odd number of memory batches
        load(B[1], k + U);
        forward_mem(B[0], k);
        B[0] = B[1];
        k += U;
    }
    // k == A: the LL + RL on-chip levels, fully unrolled, B[b & 1] is a compile-time choice
    constexpr int NBC = (LL + RL) / U;
#pragma unroll
    for (int b = 0; b < NBC; ++b) {
        // a fence per batch: without it the scheduler may hoist the loads of several unrolled batches to the top of the
        // straight-line region (registers), which is what made more than 32-48 register levels slower in round 1
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < NBC) load(B[(b + 1) & 1], A + (b + 1) * U);
        const FB& c = B[b & 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = b * U + u;  // level A + l
            level(c.a[u], c.d[u], c.s[u], c.r[u]);
            __builtin_nontemporal_store(sp, p_sup + (int64_t)(A + l) * sup.sk);
            __builtin_nontemporal_store(rp, p_rhs + (int64_t)(A + l) * rhs.sk);
            if (l < LL) {
                lds[(l * 2 + 0) * 64 + lane] = sp;
                lds[(l * 2 + 1) * 64 + lane] = rp;
            } else {
                S[l - LL] = sp;
                R[l - LL] = rp;
            }
        }
    }

    // ---- BACKWARD ---------------------------------------------------------------------------------
    auto loadb = [&](BB& b, int kb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            b.s[u] = p_sup[(int64_t)(kb - u) * sup.sk];
            b.r[u] = p_rhs[(int64_t)(kb - u) * rhs.sk];
        }
    };
    T o;
    auto backward_mem = [&](const BB& b, int kb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            o = b.r[u] - (b.s[u] * o);
            __builtin_nontemporal_store(o, p_out + (int64_t)(kb - u) * out.sk);
        }
    };
    BB C[2];
    int kb = A - 1;
    if (kb - U + 1 >= 0) loadb(C[0], kb);  // in flight during the on-chip part of the sweep
    o = R[RL - 1];  // out[K-1] = rhs'[K-1]
    __builtin_nontemporal_store(o, p_out + (int64_t)(dK - 1) * out.sk);
#pragma unroll
    for (int l = RL - 2; l >= 0; --l) {
        o = R[l] - (S[l] * o);
        __builtin_nontemporal_store(o, p_out + (int64_t)(A + LL + l) * out.sk);
    }
#pragma unroll 4
    for (int l = LL - 1; l >= 0; --l) {
        const T s = lds[(l * 2 + 0) * 64 + lane], r = lds[(l * 2 + 1) * 64 + lane];
        o = r - (s * o);
        __builtin_nontemporal_store(o, p_out + (int64_t)(A + l) * out.sk);
    }
    // invariant: C[0] holds the batch kb, kb - 1, ... whenever a whole batch is left
    while (kb - 2 * U + 1 >= 0) {
        loadb(C[1], kb - U);
        backward_mem(C[0], kb);
        if (kb - 3 * U + 1 >= 0) loadb(C[0], kb - 2 * U);
        backward_mem(C[1], kb - U);
        kb -= 2 * U;
    }
    if (kb - U + 1 >= 0) {
        backward_mem(C[0], kb);
        kb -= U;
    }
    {  // the `head` levels at the bottom, loaded together
        BB t;
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u <= kb) {
                t.s[u] = p_sup[(int64_t)(kb - u) * sup.sk];
                t.r[u] = p_rhs[(int64_t)(kb - u) * rhs.sk];
            }
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u <= kb) {
                o = t.r[u] - (t.s[u] * o);
                __builtin_nontemporal_store(o, p_out + (int64_t)(kb - u) * out.sk);
            }
    }
}

}  // namespace gt4mi
