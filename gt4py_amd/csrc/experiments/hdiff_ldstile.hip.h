// Horizontal diffusion with an LDS-staged `in` block -- the design BASELINE.json's north_star names ("LDS staging of
// (tile+halo) blocks"), measured once against the register-only J-march of hdiff_jmarch.hip.h (round 6).
// Tooling: reachable only from experiments/microbench.hip (section `hdiffl`).
//
// A workgroup of NW waves owns ONE I strip of 64 lanes x VEC columns (same lane -> column map and the same DPP neighbour
// exchange as the J-march: lanes 0 and 63 are halo lanes, consecutive strips overlap by two lanes) and marches down a segment
// of `seg_rows` rows of J at one K level in chunks of R = NW * RW rows, wave w producing rows [w * RW, (w + 1) * RW) of a chunk.
// Every row of `in` the segment needs -- rows -2 .. seg_rows + 1 relative to its start -- is brought from memory exactly ONCE
// per workgroup into a ring of NSLOT = 2 R + 4 row slots in LDS (a slot = 64 lanes x 16 B = 1 KiB, lane-linear, which is the
// image a `global_load_lds_dwordx4` writes and what `ds_read_b128` reads without bank conflicts).  While the waves compute
// chunk n out of the ring (R + 4 live rows: two halo rows either side), the R rows chunk n + 1 adds are in flight:
//   GLDS = true   straight to LDS (`global_load_lds_dwordx4`, no staging registers),
//   GLDS = false  global_load -> registers at the top of the chunk, ds_write_b128 at its end.
// One workgroup barrier per chunk.  lap / flx / fly stay in registers exactly as in the J-march (a wave recomputes the lap of
// its two halo rows and one fly row per chunk); `coeff` never touches LDS: it is read once, nontemporal, at the top of its chunk.
//
// Traffic model: `in` is fetched (seg_rows + 4) / seg_rows x 64 / 62 times (1.065 at 128 rows, 1.05 at 256) against the
// 1.37 measured for the J-march (whose halo rows are re-read by neighbouring strips from L2 / Infinity Cache).
#pragma once

#include <type_traits>

#include "common.hip.h"
#include "hdiff.hip.h"
#include "lane_shift.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

template <typename T, typename W, int VEC, int RW, int NW, bool GLDS>
__global__ void __launch_bounds__(NW * 64, (NW * RW >= 32) ? 2 : 3)
hdiff_ldstile_kernel(View<const T> in, View<T> out, View<const T> cf, int dI, int dJ, unsigned waves_i, unsigned segs_j,
                     int seg_rows, int xcd_group) {
    static_assert(VEC * sizeof(T) == 16, "16-byte lanes: the LDS row slot is 64 x 16 B");
    constexpr int R = NW * RW;
    constexpr int NSLOT = 2 * R + 4;
    constexpr int SLOT_BYTES = 64 * 16;
    using V = typename VecT<T, VEC>::type;
    __shared__ __attribute__((aligned(16))) char ring[NSLOT * SLOT_BYTES];

    unsigned wg = blockIdx.x;
    if (xcd_group > 0) {
        // runs of `xcd_group` consecutive segments share an XCD (the hardware deals workgroups round-robin over 8 XCDs)
        const unsigned g = (unsigned)xcd_group, round = 8u * g;
        if (wg < (gridDim.x / round) * round) {
            const unsigned q = wg / round, r = wg % round;
            wg = q * round + (r % 8u) * g + (r / 8u);
        }
    }
    const unsigned seg = wg % segs_j, column = wg / segs_j;
    const unsigned wi = column % waves_i, k = column / waves_i;
    const unsigned lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = ((int)(wi * 62u) - 1 + (int)lane) * VEC;
    const int js = (int)seg * seg_rows;
    const int seglen = (dJ - js < seg_rows) ? (dJ - js) : seg_rows;

    // Lane classes for the `in` vector: FULL (own 16-byte vector, inside the readable columns [-2, dI + 2)), OUTSIDE (nothing
    // of it is readable and nothing it computes is stored: it loads the strip's first full vector instead -- a valid aligned
    // address), STRADDLE (partly readable: the first lane of the first strip when VEC = 4, the lane that holds column dI + 1
    // unless dI + 2 is a multiple of VEC).  Only STRADDLE lanes need element-wise, register-staged loads; a strip without
    // one takes the EDGE = false instantiation of the chunk body, which has no per-lane load paths at all.
    const bool in_whole = (col >= -2) && (col + VEC <= dI + 2);
    const bool in_none = (col + VEC <= -2) || (col >= dI + 2);
    const bool straddle = !in_whole && !in_none;
    const bool is_out_lane = (lane >= 1u) && (lane < 63u);
    const bool out_whole = is_out_lane && (col >= 0) && (col + VEC <= dI);
    const bool out_some = is_out_lane && (col + VEC > 0) && (col < dI) && !out_whole;
    const bool edge_strip = __builtin_amdgcn_ballot_w64(straddle || out_some) != 0ull;

    // wave-uniform row bases (scalar registers) + one unsigned 32-bit BYTE offset per lane; the bases are biased by VEC items so
    // that the offset of the leftmost lane (col = -VEC) is not negative
    const T* __restrict__ ip = in.p + (int64_t)k * in.sk + (int64_t)js * in.sj - VEC;
    T* __restrict__ op = out.p + (int64_t)k * out.sk + (int64_t)js * out.sj - VEC;
    const T* __restrict__ cp = cf.p + (int64_t)k * cf.sk + (int64_t)js * cf.sj - VEC;
    const unsigned ucol = (unsigned)(col + VEC) * (unsigned)sizeof(T);
    // OUTSIDE lanes read the vector of the strip's lane 1 (column wi * 62 * VEC >= 0: always readable when the strip exists)
    const unsigned ucol_in = in_none ? (unsigned)((int)(wi * 62u) * VEC + VEC) * (unsigned)sizeof(T) : ucol;
    const unsigned ucol_cf = out_whole ? ucol : (unsigned)((int)(wi * 62u) * VEC + VEC) * (unsigned)sizeof(T);
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
        eoff[e] = (unsigned)(ce + VEC) * (unsigned)sizeof(T);
        int cc = col + e;
        cc = cc < 0 ? 0 : (cc > dI - 1 ? dI - 1 : cc);
        coff[e] = (unsigned)(cc + VEC) * (unsigned)sizeof(T);
    }

    // relative row r (-2 .. seglen + 1) lives in slot (r + 2) mod NSLOT; `sbase` = slot of row base - 2 (the chunk's first live
    // row), so a row of the live or the incoming block is at most one wrap away
    int sbase = 0;
    auto lds_slot = [&](int r, int base) -> char* {
        int s = sbase + (r - base + 2);
        if (s >= NSLOT) s -= NSLOT;
        return ring + s * SLOT_BYTES;
    };
    // bring relative row r of the chunk at `base` in.  GLDS: straight to LDS (STRADDLE lanes of an edge strip into `v`);
    // otherwise into `v`.
    auto fetch = [&](auto edge, int r, int base, T (&v)[VEC]) {
        constexpr bool EDGE = decltype(edge)::value;
        const T* p = ip + (int64_t)r * in.sj;
        if constexpr (GLDS) {
            if (!EDGE || !straddle) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)at(p, ucol_in),
                                                 (__attribute__((address_space(3))) void*)lds_slot(r, base), 16, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = *at(p, eoff[e]);
            }
        } else {
            if (!EDGE || !straddle) {
                vload<T, VEC>(at(p, ucol_in), v);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = *at(p, eoff[e]);
            }
        }
    };
    auto stash = [&](auto edge, int r, int base, const T (&v)[VEC]) {
        constexpr bool EDGE = decltype(edge)::value;
        if constexpr (GLDS && !EDGE) return;
        if (GLDS && !straddle) return;
        V pack;
#pragma unroll
        for (int e = 0; e < VEC; ++e) pack[e] = v[e];
        *reinterpret_cast<V*>(lds_slot(r, base) + lane * 16) = pack;
    };
    auto read_row = [&](int r, int base, T (&v)[VEC]) {
        const V pack = *reinterpret_cast<const V*>(lds_slot(r, base) + lane * 16);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = pack[e];
    };
    auto load_cf = [&](auto edge, int r, T (&v)[VEC]) {
        constexpr bool EDGE = decltype(edge)::value;
        const T* p = cp + (int64_t)r * cf.sj;
        if (!EDGE || !out_some) {
            const V pack = __builtin_nontemporal_load(reinterpret_cast<const V*>(at(p, ucol_cf)));
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = pack[e];
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = *at(p, coff[e]);
        }
    };
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

    const int last_row = seglen + 2;  // rows [-2, last_row) exist for this segment

    // One chunk for this wave.  FULL: the wave has all RW rows to produce and the next chunk's RW rows to fetch either all
    // exist (FETCH) or none does -- no per-row conditions, everything unrolled.  Otherwise (`full` false) every row is guarded.
    auto chunk = [&](auto full, auto fetch_next, auto edge, int base) {
        constexpr bool FULL = decltype(full)::value, FETCH = decltype(fetch_next)::value;
        const int r0 = base + w * RW;  // this wave's first row in the chunk
        const int nrows = FULL ? RW : ((seglen - r0 < RW) ? (seglen - r0) : RW);
        const int nb = base + R;
        T qn[RW][VEC];
        T qc[RW][VEC];  // this chunk's coeff rows: read once, nontemporal, first needed at the end of the first step
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const int r = nb + 2 + w * RW + t;
            if (FULL ? FETCH : (r < last_row)) fetch(edge, r, base, qn[t]);
        }
#pragma unroll
        for (int t = 0; t < RW; ++t)
            if (FULL || t < nrows) load_cf(edge, r0 + t, qc[t]);

        if (FULL || nrows > 0) {
            T a[VEC], bm[VEC], b[VEC], c[VEC];
            read_row(r0 - 2, base, a);
            read_row(r0 - 1, base, bm);
            read_row(r0, base, b);
            read_row(r0 + 1, base, c);
            W lap_m[VEC], lap_b[VEC], fly_prev[VEC];
            T unused, b_next_first;
            lap_row(a, bm, b, lap_m, unused);
            lap_row(bm, b, c, lap_b, b_next_first);
#pragma unroll
            for (int e = 0; e < VEC; ++e) fly_prev[e] = hd_flux<T, W, true>(lap_b[e], lap_m[e], b[e], bm[e]);
#pragma unroll
            for (int t = 0; t < RW; ++t) {
                if (FULL || t < nrows) {
                    T d[VEC];
                    read_row(r0 + t + 2, base, d);
                    W lap_c[VEC];
                    T c_next_first;
                    lap_row(b, c, d, lap_c, c_next_first);
                    W flx[VEC], fly[VEC];
                    const W lapb_next_first = lane_shift<W, false>(lap_b[0]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const W l1 = (e == VEC - 1) ? lapb_next_first : lap_b[e + 1];
                        const T i1 = (e == VEC - 1) ? b_next_first : b[e + 1];
                        flx[e] = hd_flux<T, W, true>(l1, lap_b[e], i1, b[e]);
                    }
                    const W flx_prev_last = lane_shift<W, true>(flx[VEC - 1]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) fly[e] = hd_flux<T, W, true>(lap_c[e], lap_b[e], c[e], b[e]);
                    T res[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const W fm = (e == 0) ? flx_prev_last : flx[e - 1];
                        res[e] = hd_out<T, W, W>(b[e], (W)qc[t][e], flx[e], fm, fly[e], fly_prev[e]);
                    }
                    T* orow = op + (int64_t)(r0 + t) * out.sj;
                    if (out_whole) {
                        vstore<T, VEC, true>(at(orow, ucol), res);
                    } else if (decltype(edge)::value && out_some) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (col + e >= 0 && col + e < dI) at(orow, ucol)[e] = res[e];
                    }
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        b[e] = c[e];
                        c[e] = d[e];
                        lap_b[e] = lap_c[e];
                        fly_prev[e] = fly[e];
                    }
                    b_next_first = c_next_first;
                }
            }
        }
        // ---- hand over: the register-staged rows into the ring -------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const int r = nb + 2 + w * RW + t;
            if (FULL ? FETCH : (r < last_row)) stash(edge, r, base, qn[t]);
        }
    };

    // ---- prologue: rows -2 .. 1 (one per wave, round-robin) and the first chunk's R rows 2 .. R + 1 ------------------------
    {
        constexpr std::true_type edge{};
        T head[(NW >= 4) ? 1 : 4][VEC], first[RW][VEC];
        int h = 0;
        for (int r = -2 + w; r < 2; r += NW, ++h) fetch(edge, r, 0, head[h]);
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const int r = 2 + w * RW + t;
            if (r < last_row) fetch(edge, r, 0, first[t]);
        }
        h = 0;
        for (int r = -2 + w; r < 2; r += NW, ++h) stash(edge, r, 0, head[h]);
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const int r = 2 + w * RW + t;
            if (r < last_row) stash(edge, r, 0, first[t]);
        }
    }
    __syncthreads();

    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    for (int base = 0; base < seglen; base += R) {
        // (workgroup-uniform) every wave has RW rows to produce; the next chunk's rows all exist or none does
        const bool all_rows = base + R <= seglen;
        const bool next_all = base + 2 * R + 2 <= last_row, next_none = base + R + 2 >= last_row;
        if (all_rows && next_all) {
            if (edge_strip) chunk(yes, yes, yes, base);
            else chunk(yes, yes, no, base);
        } else if (all_rows && next_none) {
            if (edge_strip) chunk(yes, no, yes, base);
            else chunk(yes, no, no, base);
        } else {
            chunk(no, no, yes, base);
        }
        sbase += R;
        if (sbase >= NSLOT) sbase -= NSLOT;
        __syncthreads();
    }
}

template <typename T, typename W, int VEC, int RW, int NW, bool GLDS>
inline void hdiff_ldstile_launch(const View<const T>& in, const View<T>& out, const View<const T>& cf, int dI, int dJ, int dK,
                                 int seg_rows, int xcd_group, hipStream_t stream) {
    const unsigned waves_i = (unsigned)cdiv(dI, 62 * VEC);
    const unsigned segs_j = (unsigned)cdiv(dJ, seg_rows);
    const unsigned nb = waves_i * segs_j * (unsigned)dK;
    hipLaunchKernelGGL((hdiff_ldstile_kernel<T, W, VEC, RW, NW, GLDS>), dim3(nb), dim3(NW * 64), 0, stream, in, out, cf, dI, dJ,
                       waves_i, segs_j, seg_rows, xcd_group);
}

// The second variant of round 6 -- only the rows neighbouring waves SHARE go through LDS -- won and is the product kernel:
// ../hdiff_share.hip.h.

}  // namespace gt4mi
