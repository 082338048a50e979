// Horizontal diffusion, register J-march with the halo rows of neighbouring waves exchanged through LDS (the fast path for
// I-contiguous fields with 16-byte lanes since round 6).
//
// Lane -> column map and horizontal neighbours exactly as in hdiff_jmarch.hip.h: a wave owns a strip of 64 lanes x VEC columns
// (VEC contiguous elements = one 16-byte vector per lane), lanes 0 and 63 are halo lanes, consecutive strips overlap by two
// lanes, the i +- 1 neighbours come from the adjacent lane with DPP wave shifts.  What changed is where the j +- 1, +- 2 rows come
// from.  A workgroup is NW waves on NW consecutive blocks of LJ rows of one strip; every wave loads ITS OWN LJ rows of `in`
// (all in flight at once, kept in registers), writes the first two and the last two of them to LDS (4 KiB per wave), the
// workgroup meets at ONE barrier, and each wave picks the two rows above and the two rows below its block out of its
// neighbours' slots.  Only the workgroup's outer halo -- two rows above wave 0, two below wave NW - 1 -- is loaded from memory
// a second time on the chip.  Row loads per workgroup: NW * LJ + 4 instead of NW * (LJ + 4) (16 + 4 instead of 32 at 4 x 4).
//
// Measured against the J-march on MI355X (profiles/r6_hdiff_lds_tile.txt, every variant bit-identical to the one-thread-per-
// point kernel): float32 1024 x 1024 x 80  0.1774 -> 0.1690 ms, float64 512 x 1024 x 80  0.1759 -> 0.1690 ms; memory-side traffic
// 1.125-1.133 x -> 1.05-1.06 x of the algorithmic bytes with XCD runs of 4, 1.08-1.095 x with the runs of 2 the library uses (same
// speed or 0.5-2 % faster on every box; the difference is halo rows served by the Infinity Cache).  In the product's call path with
// the fields placed by role, alternating processes on one box (profiles/r6_hdiff_share_ab_product_path.log): float32 0.1699 -> 0.1664 ms
// (0.740 -> 0.756 of the HBM peak), float64 0.1648 -> 0.1628 (0.764 -> 0.773); XCD runs of 4 there: +0.7 % / +2.1 % time.
// The full LDS ring (every row of `in` staged through LDS, double-buffered along J, `global_load_lds_dwordx4` or register-staged)
// moves 1.10-1.12 x and is 8 % (float64) to 30 % (float32) SLOWER: one barrier per chunk that also drains the chunk's stores, and
// 2-3 waves per SIMD under 36-68 KiB of LDS.
//
// lap / flx / fly are computed once per point of a wave's block plus the lap of its two halo rows and one fly row, in the same
// arithmetic as hdiff_generic_kernel (no FMA, no reassociation).
#pragma once

#include <type_traits>

#include "common.hip.h"
#include "lane_shift.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

// `lead`: the views' origins lie that many items past a 16-byte boundary (all three alike, hdiff_common_lead): the lanes start
// `lead` columns further left, which makes every lane's vector naturally aligned.
template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC, int LJ, int NW, int XCDG, int MINW>
__global__ void __launch_bounds__(NW * 64, MINW)
hdiff_share_kernel(View<const T> in, View<T> out, View<const T> cf, PW coeff_scalar, int dI, int dJ, unsigned waves_i,
                   unsigned groups_j, int lead) {
    static_assert(VEC * sizeof(T) == 16 && LJ >= 4, "16-byte lanes; a wave publishes two rows at either end of its block");
    using V = typename VecT<T, VEC>::type;
    __shared__ __attribute__((aligned(16))) char shared_rows[NW * 4 * 1024];

    // workgroups ordered along J, then I, then K; runs of XCDG of them share an XCD (see lap5.hip.h)
    unsigned wg = blockIdx.x;
    if constexpr (XCDG > 0) wg = xcd_remap_grouped<(unsigned)XCDG>(wg, gridDim.x);
    const unsigned jg = wg % groups_j, column = wg / groups_j;
    const unsigned wi = column % waves_i, k = column / waves_i;
    const unsigned lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = ((int)(wi * 62u) - 1 + (int)lane) * VEC - lead;  // first column of this lane
    const int j0 = ((int)jg * NW + w) * LJ;                           // first row of this wave
    const int nrows = (dJ - j0 < LJ) ? (dJ - j0) : LJ;                // (<= 0 for the idle waves of the last workgroup)

    // Lane classes of the `in` vector against the readable columns [-2, dI + 2): WHOLE (inside), NONE (nothing of it is
    // readable and nothing the lane computes is stored: it loads a vector that certainly is readable instead, so that the
    // common path has no per-lane condition), STRADDLE (element by element at clamped columns).  Same for coeff / out against
    // [0, dI).  A strip without STRADDLE / partial-output lanes takes the EDGE = false instantiation of the body.
    const bool in_whole = (col >= -2) && (col + VEC <= dI + 2);
    const bool in_none = (col + VEC <= -2) || (col >= dI + 2);
    const bool straddle = !in_whole && !in_none;
    const bool is_out_lane = (lane >= 1u) && (lane < 63u);
    const bool out_whole = is_out_lane && (col >= 0) && (col + VEC <= dI);
    const bool out_some = is_out_lane && (col + VEC > 0) && (col < dI) && !out_whole;
    const bool edge_strip = __builtin_amdgcn_ballot_w64(straddle || out_some) != 0ull;

    // wave-uniform row bases + one unsigned 32-bit BYTE offset per lane; the bases are biased by 2 VEC items so that the offset
    // of the leftmost lane (col = -VEC - lead > -2 VEC) is not negative
    constexpr int BIAS = 2 * VEC;
    const T* __restrict__ ip = in.p + (int64_t)k * in.sk - BIAS;
    T* __restrict__ op = out.p + (int64_t)k * out.sk - BIAS;
    const T* __restrict__ cp = COEFF_FIELD ? (cf.p + (int64_t)k * cf.sk - BIAS) : nullptr;
    auto off = [](int c) { return (unsigned)(c + BIAS) * (unsigned)sizeof(T); };
    // the first lane-aligned vector at or right of column 0 (columns VEC' .. VEC' + VEC - 1 with VEC' = (VEC - lead) % VEC):
    // readable and inside [0, dI) for every domain the launcher sends here (dI >= 32)
    const int safe_col = (VEC - lead) % VEC;
    const unsigned ucol = off(col);
    const unsigned ucol_in = in_none ? off(safe_col) : ucol;
    const unsigned ucol_cf = out_whole ? ucol : off(safe_col);
    auto at = [](auto* base, unsigned byte_off) {
        using P = decltype(base);
        if constexpr (std::is_const_v<std::remove_pointer_t<P>>) return (P)((const char*)base + byte_off);
        else return (P)((char*)base + byte_off);
    };
    unsigned eoff[VEC], coff[VEC];  // (EDGE) element-wise byte offsets at clamped columns -- always valid addresses
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        int ce = col + e;
        ce = ce < -2 ? -2 : (ce > dI + 1 ? dI + 1 : ce);
        eoff[e] = off(ce);
        int cc = col + e;
        cc = cc < 0 ? 0 : (cc > dI - 1 ? dI - 1 : cc);
        coff[e] = off(cc);
    }
    // `in` row j (rows -2 .. dJ + 1 are readable; a row past the end is clamped to the last one: valid address, unused value)
    auto load_in = [&](auto edge, int j, T (&v)[VEC]) {
        const int jc = j > dJ + 1 ? dJ + 1 : j;
        const T* p = ip + (int64_t)jc * in.sj;
        if (!decltype(edge)::value || !straddle) {
            vload<T, VEC>(at(p, ucol_in), v);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = *at(p, eoff[e]);
        }
    };
    // `coeff` is read exactly once: nontemporal (round 5, profiles/r5_nt_loads_column_kernels.txt)
    auto load_cf = [&](auto edge, int j, T (&v)[VEC]) {
        const int jc = j > dJ - 1 ? dJ - 1 : j;
        const T* p = cp + (int64_t)jc * cf.sj;
        if (!decltype(edge)::value || !out_some) {
            const V pack = __builtin_nontemporal_load(reinterpret_cast<const V*>(at(p, ucol_cf)));
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = pack[e];
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = *at(p, coff[e]);
        }
    };
    auto publish = [&](int wave, int slot, const T (&v)[VEC]) {
        V pack;
#pragma unroll
        for (int e = 0; e < VEC; ++e) pack[e] = v[e];
        *reinterpret_cast<V*>(shared_rows + (wave * 4 + slot) * 1024 + lane * 16) = pack;
    };
    auto pick = [&](int wave, int slot, T (&v)[VEC]) {
        const V pack = *reinterpret_cast<const V*>(shared_rows + (wave * 4 + slot) * 1024 + lane * 16);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = pack[e];
    };
    // lap of row c given the rows below (b) and above (d); also hands back the +i neighbour of the lane's last column
    auto lap_row = [&](const T (&b)[VEC], const T (&c)[VEC], const T (&d)[VEC], W (&lap)[VEC], T& c_next_first) {
        const T c_prev_last = lane_shift<T, true>(c[VEC - 1]);
        c_next_first = lane_shift<T, false>(c[0]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const T im = (e == 0) ? c_prev_last : c[e - 1];
            const T ipv = (e == VEC - 1) ? c_next_first : c[e + 1];
            lap[e] = hd_lap<T, W>(c[e], ipv, im, d[e], b[e]);
        }
    };

    auto body = [&](auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        // rows[t] = `in` row j0 - 2 + t, t = 0 .. LJ + 3
        T rows[LJ + 4][VEC], qc[COEFF_FIELD ? LJ : 1][VEC];
        // the rows the neighbours wait for go first (loads return in order)
        load_in(edge, j0, rows[2]);
        load_in(edge, j0 + 1, rows[3]);
        load_in(edge, j0 + LJ - 2, rows[LJ]);
        load_in(edge, j0 + LJ - 1, rows[LJ + 1]);
#pragma unroll
        for (int t = 2; t < LJ - 2; ++t) load_in(edge, j0 + t, rows[2 + t]);
        if (w == 0) {  // the workgroup's outer halo
            load_in(edge, j0 - 2, rows[0]);
            load_in(edge, j0 - 1, rows[1]);
        }
        if (w == NW - 1) {
            load_in(edge, j0 + LJ, rows[LJ + 2]);
            load_in(edge, j0 + LJ + 1, rows[LJ + 3]);
        }
        if constexpr (COEFF_FIELD) {
#pragma unroll
            for (int t = 0; t < LJ; ++t) load_cf(edge, j0 + t, qc[t]);
        }
        publish(w, 0, rows[2]);
        publish(w, 1, rows[3]);
        publish(w, 2, rows[LJ]);
        publish(w, 3, rows[LJ + 1]);
        __syncthreads();
        if (w > 0) {
            pick(w - 1, 2, rows[0]);
            pick(w - 1, 3, rows[1]);
        }
        if (w < NW - 1) {
            pick(w + 1, 0, rows[LJ + 2]);
            pick(w + 1, 1, rows[LJ + 3]);
        }
        if (nrows <= 0) return;
        W lap_m[VEC], lap_b[VEC], fly_prev[VEC];
        T unused, b_next_first;
        lap_row(rows[0], rows[1], rows[2], lap_m, unused);
        lap_row(rows[1], rows[2], rows[3], lap_b, b_next_first);
#pragma unroll
        for (int e = 0; e < VEC; ++e) fly_prev[e] = hd_flux<T, W, LIMITER>(lap_b[e], lap_m[e], rows[2][e], rows[1][e]);
#pragma unroll
        for (int t = 0; t < LJ; ++t) {
            if (t < nrows) {
                const T(&b)[VEC] = rows[t + 2];
                const T(&c)[VEC] = rows[t + 3];
                const T(&d)[VEC] = rows[t + 4];
                W lap_c[VEC];
                T c_next_first;
                lap_row(b, c, d, lap_c, c_next_first);
                // flx(row j) at column e needs lap_b and `in` row j at column e + 1
                W flx[VEC], fly[VEC];
                const W lapb_next_first = lane_shift<W, false>(lap_b[0]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const W l1 = (e == VEC - 1) ? lapb_next_first : lap_b[e + 1];
                    const T i1 = (e == VEC - 1) ? b_next_first : b[e + 1];
                    flx[e] = hd_flux<T, W, LIMITER>(l1, lap_b[e], i1, b[e]);
                }
                const W flx_prev_last = lane_shift<W, true>(flx[VEC - 1]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) fly[e] = hd_flux<T, W, LIMITER>(lap_c[e], lap_b[e], c[e], b[e]);
                T res[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const W fm = (e == 0) ? flx_prev_last : flx[e - 1];
                    PW cv;
                    if constexpr (COEFF_FIELD) cv = (PW)qc[t][e];
                    else cv = coeff_scalar;
                    res[e] = hd_out<T, W, PW>(b[e], cv, flx[e], fm, fly[e], fly_prev[e]);
                }
                T* orow = op + (int64_t)(j0 + t) * out.sj;
                if (out_whole) {
                    vstore<T, VEC, true>(at(orow, ucol), res);
                } else if (EDGE && out_some) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (col + e >= 0 && col + e < dI) at(orow, ucol)[e] = res[e];
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    lap_b[e] = lap_c[e];
                    fly_prev[e] = fly[e];
                }
                b_next_first = c_next_first;
            }
        }
    };
    if (edge_strip) body(std::true_type{});
    else body(std::false_type{});
}

// Rows per wave / waves per workgroup (MI355X; profiles/r6_hdiff_lds_tile.txt).  4 x 4 is the fastest shape for both BASELINE
// configurations AND the one with the least memory-side traffic (1.05-1.06 x; 6 rows x 4 waves: 1.12 x, 4 rows x 8 waves: 1.07-1.09 x);
// 8, 12 and 16 waves lose 5-30 % for float32 fields with float64 internals (128 registers there: the barrier then spans most of a CU).
template <typename T>
struct HdiffShareTuning {
    static constexpr int LJ = 4;
    static constexpr int NW = 4;
    // workgroups per XCD run: 2 (1, 2 and no remap are within 0.3 %; 4 -- the J-march's value -- is 0.5-2 % behind, 8 and 16 2-4 %)
    static constexpr int XCDG = 2;
    static constexpr int XCDG_ALT = 4;
    static constexpr int MINW = 4;
};

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC, int LJ, int NW, int XCDG, int MINW>
inline int hdiff_launch_share_shape(const View<const T>& in, const View<T>& out, const View<const T>& cf, PW coeff_scalar,
                                    const int64_t d[3], hipStream_t stream, int lead) {
    const unsigned waves_i = (unsigned)cdiv(d[0] + lead, (int64_t)62 * VEC);
    const unsigned groups_j = (unsigned)cdiv(d[1], (int64_t)LJ * NW);
    const int64_t nblocks = (int64_t)waves_i * groups_j * d[2];
    if (nblocks > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "hdiff: domain too large for one launch");
    // launch_dynamic_lds(): the throttle of the overlapped distributed apply -- bytes per workgroup such that at most N workgroups
    // share a CU (common.hip.h lds_for_workgroups_per_cu).  This kernel has static LDS of its own: only the rest is requested.
    constexpr unsigned own_lds = NW * 4 * 1024;
    const unsigned throttle = launch_dynamic_lds() > own_lds ? launch_dynamic_lds() - own_lds : 0u;
    hipLaunchKernelGGL((hdiff_share_kernel<T, W, PW, LIMITER, COEFF_FIELD, VEC, LJ, NW, XCDG, MINW>), dim3((unsigned)nblocks),
                       dim3(NW * 64), throttle, stream, in, out, cf, coeff_scalar, (int)d[0], (int)d[1], waves_i, groups_j, lead);
    return GT4MI_OK;
}

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC>
inline int hdiff_launch_share(const View<const T>& in, const View<T>& out, const View<const T>& cf, PW coeff_scalar,
                              const int64_t d[3], hipStream_t stream, int lead) {
    using Tu = HdiffShareTuning<T>;
    // GT4MI_HDIFF_SHARE_XCD=<runs>: the other measured XCD run length, for A/B runs on ONE box in the product's call path
    static const int runs = env_int("GT4MI_HDIFF_SHARE_XCD", Tu::XCDG);
    if (runs != Tu::XCDG)
        return hdiff_launch_share_shape<T, W, PW, LIMITER, COEFF_FIELD, VEC, Tu::LJ, Tu::NW, Tu::XCDG_ALT, Tu::MINW>(in, out, cf, coeff_scalar, d,
                                                                                                                 stream, lead);
    return hdiff_launch_share_shape<T, W, PW, LIMITER, COEFF_FIELD, VEC, Tu::LJ, Tu::NW, Tu::XCDG, Tu::MINW>(in, out, cf, coeff_scalar, d,
                                                                                                         stream, lead);
}

}  // namespace gt4mi
