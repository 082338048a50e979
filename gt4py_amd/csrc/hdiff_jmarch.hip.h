// Horizontal diffusion, wave-autonomous "J-march" kernel for I-contiguous fields: the fast path of rounds 1-5.  Since round 6
// whole domains on 16-byte lanes run hdiff_share.hip.h (the same lane map; the waves of a workgroup exchange their halo rows
// through LDS instead of each loading them); this kernel remains the path for 8- / 4-byte lanes, the strips of the boundary
// ring (hdiff_ring.hip.h) and GT4MI_HDIFF_SHARE=0.
//
// One wave owns a strip of 64*VEC columns (VEC contiguous elements per lane = one 8/16-byte vector)
// and walks down LJ rows of J at one K level.  Per step it loads ONE new row of `in` (row j+2),
// and keeps in registers: in rows j, j+1, lap rows j, j+1 and fly row j-1.  Horizontal neighbours
// come from the adjacent lane with DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no
// LDS, no barriers.  The first and last H lanes of a wave are halo lanes (they only feed their
// neighbours), so consecutive waves overlap by 2*H lanes (H = 1 for VEC >= 2, 2 for VEC == 1):
//
//   lane:      0      1 ..................... 62      63
//   columns: [halo][ ---- 62*VEC outputs ---- ][halo]      wave w starts at (w*62 - 1)*VEC
//
// lap, flx, fly are computed exactly once per point inside a strip (plus the halo lanes and the
// 2-row prologue), in the same arithmetic as hdiff_generic_kernel.
#pragma once

#include "common.hip.h"
#include "lane_shift.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

inline bool hdiff_jmarch_enabled() {
    static const bool on = [] {
        const char* e = getenv("GT4MI_HDIFF_GENERIC");
        return !(e && e[0] == '1');
    }();
    return on;
}

// One strip: wave `wi` along I, rows [tj * LJ, tj * LJ + LJ) of level k, of a domain of dI x dJ points whose origin the
// views point at.  Shared by the whole-domain kernel below and by the boundary-ring kernel (hdiff_ring.hip.h).
// OPT (bit mask; value-identical variants that only change the instruction mix -- for float fields with double internals the
// kernel is VALU co-limited: 41 VALU instructions per lattice update, 7 of them conversions, profiles/r3_hdiff_stall_counters.txt):
//   1  the two aligned additions of the lap's f32 sum (+ row above, + row below) and fly's f32 difference as PACKED f32 pairs
//      (v_pk_add_f32: two IEEE additions per instruction, each rounded exactly like the scalar one)
//   2  a row's widened copy (W)in, made for its lap, is kept for the row's hd_out two steps later instead of converting again
//   4  ROLLED: the march is a rolled loop over chunks of PF rows whose body is unrolled PF times, the prefetch queue a ring that
//      is refilled in place -- register use no longer grows with LJ (fully unrolled, the scheduler hoists the loads of later steps:
//      138 registers at LJ = 8, 158 at 12, 214 at 32 against 122 at 6), so strips can be long behind a short prefetch window
constexpr int HD_OPT_PACKED = 1, HD_OPT_KEEP_WIDE = 2, HD_OPT_ROLLED = 4;
// (round 5, after the column kernels gained 5-9 % from them) nontemporal loads of `coeff` -- read exactly once -- and of `in`
constexpr int HD_OPT_NT_COEFF = 8, HD_OPT_NT_IN = 16;

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC, int LJ, int PF, int OPT = 0>
// `lead`: the views' origins lie that many items past a 16-byte boundary (all three alike): the lanes then start `lead`
// columns further left, which makes every lane's vector naturally aligned again; the lanes that straddle the edge of the
// readable / writable columns take the element-wise paths that partial vectors at the domain's edges take anyway.
__device__ __forceinline__ void hdiff_jmarch_strip(const View<const T>& in, const View<T>& out, const View<const T>& cf,
                                                   PW coeff_scalar, int dI, int dJ, unsigned wi, unsigned tj, unsigned k,
                                                   int lead = 0) {
    constexpr int H = (VEC >= 2) ? 1 : 2;   // halo lanes per side
    constexpr int OUT_LANES = 64 - 2 * H;
    const unsigned lane = threadIdx.x & 63;
    const int col = ((int)(wi * OUT_LANES) - H + (int)lane) * VEC - lead;  // first column of this lane
    const int j0 = (int)tj * LJ;
    const int nrows = (dJ - j0 < LJ) ? (dJ - j0) : LJ;

    // `in` is readable on columns [-2, dI+2), out/coeff on [0, dI).
    const bool in_full = (col >= -2) && (col + VEC <= dI + 2);
    const bool in_any = (col + VEC > -2) && (col < dI + 2);
    const bool is_out_lane = (lane >= (unsigned)H) && (lane < (unsigned)(64 - H));
    const bool out_full = is_out_lane && (col >= 0) && (col + VEC <= dI);
    const bool out_any = is_out_lane && (col + VEC > 0) && (col < dI);

    const T* __restrict__ ip = in.p + (int64_t)k * in.sk + col;
    T* __restrict__ op = out.p + (int64_t)k * out.sk + col;
    const T* __restrict__ cp = COEFF_FIELD ? (cf.p + (int64_t)k * cf.sk + col) : nullptr;

    auto vload_nt = [](const T* p, T (&r)[VEC]) {
        if constexpr (VEC == 1) {
            r[0] = __builtin_nontemporal_load(p);
        } else {
            using V = typename VecT<T, VEC>::type;
            const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#pragma unroll
            for (int e = 0; e < VEC; ++e) r[e] = v[e];
        }
    };
    auto load_in = [&](int j, T (&r)[VEC]) {
        const T* p = ip + (int64_t)j * in.sj;
        if (in_full) {
            if constexpr ((OPT & HD_OPT_NT_IN) != 0) vload_nt(p, r);
            else vload<T, VEC>(p, r);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                r[e] = (in_any && col + e >= -2 && col + e < dI + 2) ? p[e] : (T)0;
        }
    };
    auto load_cf = [&](int j, T (&r)[VEC]) {
        const T* p = cp + (int64_t)j * cf.sj;
        if (out_full) {
            if constexpr ((OPT & HD_OPT_NT_COEFF) != 0) vload_nt(p, r);
            else vload<T, VEC>(p, r);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                r[e] = (out_any && col + e >= 0 && col + e < dI) ? p[e] : (T)0;
        }
    };
    // lap of row `c` given the rows below (b) and above (d); fills right-shifted copy of c's
    // first element (the +i neighbour of the lane's last column) for reuse by the flux.
    constexpr bool PACKED = (OPT & HD_OPT_PACKED) != 0 && sizeof(T) == 4 && VEC % 2 == 0;
    constexpr bool KEEP_WIDE = (OPT & HD_OPT_KEEP_WIDE) != 0 && sizeof(T) < sizeof(W);
    typedef T pair_t __attribute__((ext_vector_type(2)));
    // (`wide`: (W)c of the centre row, for KEEP_WIDE)
    auto lap_row = [&](const T (&b)[VEC], const T (&c)[VEC], const T (&d)[VEC], W (&lap)[VEC],
                       T& c_next_first, W (&wide)[VEC]) {
        const T c_prev_last = lane_shift<T, true>(c[VEC - 1]);
        c_next_first = lane_shift<T, false>(c[0]);
        if constexpr (PACKED) {
            // sum = ((ip + im) + jp) + jm: the first addition pairs neighbours one column apart (no aligned register pairs), the
            // second and third add whole rows -- two columns per instruction
            T s[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const T im = (e == 0) ? c_prev_last : c[e - 1];
                const T ipv = (e == VEC - 1) ? c_next_first : c[e + 1];
                s[e] = ipv + im;
            }
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                pair_t t = pair_t{s[e], s[e + 1]} + pair_t{d[e], d[e + 1]};
                t = t + pair_t{b[e], b[e + 1]};
                s[e] = t.x;
                s[e + 1] = t.y;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                wide[e] = (W)c[e];
                lap[e] = ((W)4.0 * wide[e]) - (W)s[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const T im = (e == 0) ? c_prev_last : c[e - 1];
                const T ipv = (e == VEC - 1) ? c_next_first : c[e + 1];
                if constexpr (KEEP_WIDE) {
                    const T sum = ((ipv + im) + d[e]) + b[e];
                    wide[e] = (W)c[e];
                    lap[e] = ((W)4.0 * wide[e]) - (W)sum;
                } else {
                    lap[e] = hd_lap<T, W>(c[e], ipv, im, d[e], b[e]);
                }
            }
        }
    };
    auto fly_row = [&](const W (&lap_hi)[VEC], const W (&lap_lo)[VEC], const T (&in_hi)[VEC],
                       const T (&in_lo)[VEC], W (&fly)[VEC]) {
        if constexpr (PACKED && LIMITER) {
            T dd[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                const pair_t t = pair_t{in_hi[e], in_hi[e + 1]} - pair_t{in_lo[e], in_lo[e + 1]};
                dd[e] = t.x;
                dd[e + 1] = t.y;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const W res = lap_hi[e] - lap_lo[e];
                fly[e] = ((res * (W)dd[e]) > (W)0) ? (W)0 : res;
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                fly[e] = hd_flux<T, W, LIMITER>(lap_hi[e], lap_lo[e], in_hi[e], in_lo[e]);
        }
    };

    // ---- prologue: rows j0-2 .. j0+1 -> lap(j0-1), lap(j0), fly(j0-1) ---------------------------
    T a[VEC], bm[VEC], b[VEC], c[VEC];
    load_in(j0 - 2, a);
    load_in(j0 - 1, bm);
    load_in(j0, b);
    load_in(j0 + 1, c);
    // prefetch queue: rows j0+2 .. j0+1+PF
    T q[PF][VEC];
    T qc[PF][VEC];
#pragma unroll
    for (int t = 0; t < PF; ++t) {
        if (t < nrows) load_in(j0 + 2 + t, q[t]);
        if constexpr (COEFF_FIELD)
            if (t < nrows) load_cf(j0 + t, qc[t]);
    }
    W lap_m[VEC], lap_b[VEC], fly_prev[VEC];
    W wide_unused[VEC], wide_b[VEC];  // (KEEP_WIDE) (W) of row b: made by its lap, used by its hd_out
    T unused, b_next_first;
    lap_row(a, bm, b, lap_m, unused, wide_unused);
    lap_row(bm, b, c, lap_b, b_next_first, wide_b);
    fly_row(lap_b, lap_m, b, bm, fly_prev);

    constexpr bool ROLLED = (OPT & HD_OPT_ROLLED) != 0;
    static_assert(!ROLLED || LJ % PF == 0, "rolled march: whole chunks of PF rows");
    // `slot`: (ROLLED) the ring slot that holds this step's rows -- a compile-time constant inside the unrolled chunk
    auto step = [&](int jj, int nr, const int slot) {
        // row j = j0 + jj is produced; q[slot] holds in row j+2, qc[slot] holds coeff row j (slot 0 of a shifting queue, or
        // this step's slot of the ring)
        T d[VEC], cfr[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] = q[slot][e];
        if constexpr (COEFF_FIELD) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) cfr[e] = qc[slot][e];
        }
        if constexpr (!ROLLED) {
#pragma unroll
            for (int t = 0; t + 1 < PF; ++t) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    q[t][e] = q[t + 1][e];
                    if constexpr (COEFF_FIELD) qc[t][e] = qc[t + 1][e];
                }
            }
        }
        if (jj + PF < nr) {
            load_in(j0 + jj + PF + 2, q[ROLLED ? slot : PF - 1]);
            if constexpr (COEFF_FIELD) load_cf(j0 + jj + PF, qc[ROLLED ? slot : PF - 1]);
        }
        W lap_c[VEC], wide_c[VEC];
        T c_next_first;
        lap_row(b, c, d, lap_c, c_next_first, wide_c);
        // flx(row j) at column e needs lap_b and in row j at column e+1
        W flx[VEC], fly[VEC];
        const W lapb_next_first = lane_shift<W, false>(lap_b[0]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const W l1 = (e == VEC - 1) ? lapb_next_first : lap_b[e + 1];
            const T i1 = (e == VEC - 1) ? b_next_first : b[e + 1];
            flx[e] = hd_flux<T, W, LIMITER>(l1, lap_b[e], i1, b[e]);
        }
        const W flx_prev_last = lane_shift<W, true>(flx[VEC - 1]);
        fly_row(lap_c, lap_b, c, b, fly);
        T res[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const W fm = (e == 0) ? flx_prev_last : flx[e - 1];
            PW cv;
            if constexpr (COEFF_FIELD) cv = (PW)cfr[e];
            else cv = coeff_scalar;
            if constexpr (KEEP_WIDE && sizeof(PW) == sizeof(W)) {
                const W sden = ((flx[e] - fm) + fly[e]) - fly_prev[e];
                res[e] = (T)((PW)wide_b[e] - (cv * (PW)sden));  // hd_out with (PW)in0 taken from the row's lap
            } else {
                res[e] = hd_out<T, W, PW>(b[e], cv, flx[e], fm, fly[e], fly_prev[e]);
            }
        }
        T* o = op + (int64_t)(j0 + jj) * out.sj;
        if (out_full) {
            vstore<T, VEC, true>(o, res);
        } else if (out_any) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (col + e >= 0 && col + e < dI) o[e] = res[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            b[e] = c[e];
            c[e] = d[e];
            lap_b[e] = lap_c[e];
            fly_prev[e] = fly[e];
            if constexpr (KEEP_WIDE) wide_b[e] = wide_c[e];
        }
        b_next_first = c_next_first;
    };

    if constexpr (ROLLED) {
#pragma unroll 1
        for (int base = 0; base < nrows; base += PF) {
#pragma unroll
            for (int t = 0; t < PF; ++t)  // (unrolled: the ring slot t is a constant in every copy of the body)
                if (base + t < nrows) step(base + t, nrows, t);
        }
    } else if (nrows == LJ) {
#pragma unroll
        for (int jj = 0; jj < LJ; ++jj) step(jj, LJ, 0);
    } else {
        for (int jj = 0; jj < nrows; ++jj) step(jj, nrows, 0);
    }
}

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC, int LJ,
          int PF, int XCDG = 0, int OPT = 0, bool TEAM = false>
__global__ void __launch_bounds__(256)
hdiff_jmarch_kernel(View<const T> in, View<T> out, View<const T> cf, PW coeff_scalar, int dI,
                    int dJ, unsigned waves_i, unsigned tiles_j, unsigned groups_j, int lead) {
    if constexpr (TEAM) {
        // (experiment, microbench only) the four waves of a workgroup on four ADJACENT I strips of the SAME rows: no two waves
        // of a workgroup share a halo row, strips can be long with a short rolling prefetch window; workgroups ordered along
        // J (XCD-grouped), then I, then K.  groups_j = tiles_j here; waves_i is rounded up to whole teams by the launch.
        unsigned wg = blockIdx.x;
        if constexpr (XCDG > 0) wg = xcd_remap_grouped<(unsigned)XCDG>(wg, gridDim.x);
        const unsigned tj = wg % tiles_j, column = wg / tiles_j;
        const unsigned teams_i = (waves_i + 3u) / 4u;
        const unsigned wi = (column % teams_i) * 4u + (threadIdx.x >> 6), k = column / teams_i;
        if (wi >= waves_i) return;
        hdiff_jmarch_strip<T, W, PW, LIMITER, COEFF_FIELD, VEC, LJ, PF, OPT>(in, out, cf, coeff_scalar, dI, dJ, wi, tj, k, lead);
        return;
    }
    // A workgroup = 4 independent waves on 4 consecutive J strips of one I column, so 3 of the 4
    // strip boundaries (4 shared rows each) are re-read inside one CU; workgroups are ordered along
    // J, then I, then K, and runs of XCDG of them share an XCD (see lap5.hip.h).
    unsigned wg = blockIdx.x;
    unsigned jg, column;
    if constexpr (XCDG < 0) {
        // chunked: every (I column, K level) is dealt to the 8 XCDs as 8 contiguous chunks of workgroups along J
        // (the launch pads groups_j to a multiple of 8; hardware deals workgroups to XCDs round-robin in linear order)
        const unsigned padded = ((groups_j + 7u) / 8u) * 8u, per = padded / 8u;
        column = wg / padded;
        const unsigned r = wg % padded;
        jg = (r % 8u) * per + r / 8u;
        if (jg >= groups_j) return;
    } else {
        if constexpr (XCDG > 0) wg = xcd_remap_grouped<(unsigned)XCDG>(wg, gridDim.x);
        jg = wg % groups_j;
        column = wg / groups_j;
    }
    const unsigned tj = jg * 4 + (threadIdx.x >> 6);
    if (tj >= tiles_j) return;
    const unsigned wi = column % waves_i;
    const unsigned k = column / waves_i;

    hdiff_jmarch_strip<T, W, PW, LIMITER, COEFF_FIELD, VEC, LJ, PF, OPT>(in, out, cf, coeff_scalar, dI, dJ, wi, tj, k, lead);
}

// Rows per strip / rows prefetched ahead (MI355X, 1024x1024x80 f32 and 512x1024x80 f64): SHORT strips with all of their rows in
// flight win.  Round 5 tried the other direction at length (profiles/r5_microbench_hdiff_variants.log, every variant bit-identical):
// strips of 12 .. 128 rows -- rolled, behind a 3-6 row prefetch window, also with the four waves of a workgroup on adjacent I strips
// of the same rows -- are 5-55 % slower than 6-8 rows: the march of a wave is a chain of dependent load latencies, and what hides
// them is the number of independent waves, not the depth of one wave's queue.  float64, float32 with float64 internals and float32
// throughout all land within 1 % of 0.180 ms for the same 1.007 GB: the kernel sits on the ceiling of its 2-read : 1-write traffic
// mix (6.05 TB/s streaming, profiles/r3_microbench_rw_mix.log), not on VALU.
//   float64: 8 rows, all 8 in flight.   float32: 6 rows, all 6 in flight.  The float32 alternative 8 rows / 4 in flight moves 1.09x
//   instead of 1.14x of the algorithmic bytes at the memory side (4 halo rows per 8 instead of per 6) and is 1.3-2.0 % SLOWER on the
//   same box in the product's call path (profiles/r5_hdiff_f32_strip_ab.log): the extra "traffic" is Infinity-Cache hits on halo
//   rows (FETCH_SIZE counts them, MI355X_MICROARCH.md), which cost nothing -- so the faster shape stays.  GT4MI_HDIFF_F32_ROWS=8
//   selects the other one.
template <typename T>
struct HdiffTuning {
    // rows per strip / rows in flight.  fp32: 6 / 6.  fp64: 8 / 8 until round 5; with the nontemporal coeff loads 6 / 6 is 1.4 % ahead
    // (0.1735 vs 0.1759 ms on 512 x 1024 x 80, A-B x 3, microbench `hdiffnt`; 4 / 4 0.1752, 10 / 8 0.1787, 12 / 8 0.1848)
    static constexpr int LJ = 6;
    static constexpr int PF = 6;
    static constexpr int XCDG = 4;  // workgroups per XCD run (see lap5.hip.h Lap5Tuning::XCDG)
    // round 5: `coeff` is read exactly once (no halo): nontemporal loads for it -- same box A-B-A x 3, fp32 1024 x 1024 x 80
    // 0.707 -> 0.728 of the HBM peak, fp64 512 x 1024 x 80 0.707 -> 0.713; on `in`, whose halo rows neighbouring strips re-read,
    // the same hint costs 18 % (experiments/microbench.hip `hdiffnt`, profiles/r5_nt_loads_column_kernels.txt)
    static constexpr int OPT = HD_OPT_NT_COEFF;
};

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC, int LJ, int PF>
inline int hdiff_launch_jmarch_strips(const View<const T>& in, const View<T>& out, const View<const T>& cf, PW coeff_scalar,
                                      const int64_t d[3], hipStream_t stream, int lead) {
    constexpr int H = (VEC >= 2) ? 1 : 2;
    const unsigned waves_i = (unsigned)cdiv(d[0] + lead, (int64_t)(64 - 2 * H) * VEC);
    const unsigned tiles_j = (unsigned)cdiv(d[1], LJ);
    const unsigned groups_j = (unsigned)cdiv(tiles_j, 4);
    const int64_t nblocks = (int64_t)waves_i * groups_j * d[2];
    if (nblocks > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "hdiff: domain too large for one launch");
    hipLaunchKernelGGL((hdiff_jmarch_kernel<T, W, PW, LIMITER, COEFF_FIELD, VEC, LJ, PF, HdiffTuning<T>::XCDG, HdiffTuning<T>::OPT>),
                       dim3((unsigned)nblocks), dim3(256), launch_dynamic_lds(), stream, in, out, cf, coeff_scalar, (int)d[0],
                       (int)d[1], waves_i, tiles_j, groups_j, lead);
    return GT4MI_OK;
}

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC>
inline int hdiff_launch_jmarch_vec(const View<const T>& in, const View<T>& out,
                                   const View<const T>& cf, PW coeff_scalar, const int64_t d[3],
                                   hipStream_t stream, int lead = 0) {
    if constexpr (sizeof(T) == 8 && VEC == 2) {
        // GT4MI_HDIFF_F64_ROWS=8: the float64 strips of rounds 2-4 (8 rows, 8 in flight), for A/B runs on ONE box
        static const int rows64 = env_int("GT4MI_HDIFF_F64_ROWS", HdiffTuning<T>::LJ);
        if (rows64 == 8) return hdiff_launch_jmarch_strips<T, W, PW, LIMITER, COEFF_FIELD, VEC, 8, 8>(in, out, cf, coeff_scalar, d, stream, lead);
    }
    if constexpr (sizeof(T) == 4 && VEC == 4) {
        // GT4MI_HDIFF_F32_ROWS=8: float32 strips of 8 rows with 4 in flight instead of 6 / 6 -- for A/B runs on ONE box (the two
        // differ by less than boxes do: profiles/r5_hdiff_f32_strip_ab.log)
        static const int rows = env_int("GT4MI_HDIFF_F32_ROWS", HdiffTuning<T>::LJ);
        if (rows == 8) return hdiff_launch_jmarch_strips<T, W, PW, LIMITER, COEFF_FIELD, VEC, 8, 4>(in, out, cf, coeff_scalar, d, stream, lead);
    }
    return hdiff_launch_jmarch_strips<T, W, PW, LIMITER, COEFF_FIELD, VEC, HdiffTuning<T>::LJ, HdiffTuning<T>::PF>(in, out, cf, coeff_scalar, d,
                                                                                                             stream, lead);
}

// 16-byte lanes are possible when the rows of all fields are 16-byte aligned among themselves and the origins lie equally
// far (`*lead` items, 0 .. vec - 1) past a 16-byte boundary -- not only when they lie ON one: a storage allocated with the
// default aligned_index and used from origin (2, 2, 0) puts float32 fields 8 bytes off, which used to mean 4-byte lanes
// (278 instead of 410 GLUPS on 1024 x 1024 x 80, profiles/r3_misaligned_origin.log).
template <typename V>
inline bool vec_rows_ok(const V& v, int vec, int* lead) {
    if (v.si != 1 || v.sj % vec != 0 || v.sk % vec != 0) return false;
    const uintptr_t bytes = reinterpret_cast<uintptr_t>(v.p) % (vec * sizeof(*v.p));
    if (bytes % sizeof(*v.p) != 0) return false;
    *lead = (int)(bytes / sizeof(*v.p));
    return true;
}

template <typename T, bool COEFF_FIELD>
inline bool hdiff_common_lead(const View<const T>& in, const View<T>& out, const View<const T>& cf, int vec, int* lead) {
    int li = 0, lo = 0, lc = 0;
    if (!vec_rows_ok(in, vec, &li) || !vec_rows_ok(out, vec, &lo)) return false;
    if (COEFF_FIELD && !vec_rows_ok(cf, vec, &lc)) return false;
    if (li != lo || (COEFF_FIELD && lc != li)) return false;
    *lead = li;
    return true;
}

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD>
inline int hdiff_launch_jmarch(const View<const T>& in, const View<T>& out, const View<const T>& cf,
                               PW coeff_scalar, const int64_t d[3], hipStream_t stream) {
    constexpr int VMAX = 16 / sizeof(T);
    int lead = 0;
    if (hdiff_common_lead<T, COEFF_FIELD>(in, out, cf, VMAX, &lead))
        return hdiff_launch_jmarch_vec<T, W, PW, LIMITER, COEFF_FIELD, VMAX>(in, out, cf, coeff_scalar, d, stream, lead);
    return hdiff_launch_jmarch_vec<T, W, PW, LIMITER, COEFF_FIELD, 1>(in, out, cf, coeff_scalar, d, stream);
}

}  // namespace gt4mi
