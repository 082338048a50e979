// 5-point star stencils for gfx950: out[i,j,k] = f(in[i,j,k], in[i-1,j,k], in[i+1,j,k], in[i,j-1,k], in[i,j+1,k]).
//
// Reference semantics: the single-statement PARALLEL stencils listed in include/gt4py_amd.h
// (numpy backend code shape: /root/reference/src/gt4py/cartesian/gtc/numpy/npir_codegen.py:205-212).
//
// Roofline: HBM.  Algorithmic traffic 2*sizeof(T) bytes per lattice update (one read, one write).
//
// Fast path ("strip" kernel): a workgroup owns a strip BLOCK*VEC points wide in I (contiguous, one
// 16-byte vector per lane) and LJ rows tall in J at one K level.
//   1. every lane issues the loads of all LJ+2 rows of its column vector up front (maximum bytes in
//      flight per wave; measured +8 % over a rolling three-row window on MI355X),
//   2. the I-neighbours in[i-1], in[i+VEC] come from the adjacent lanes with DPP wave shifts
//      (v_mov_b32_dpp wave_shr:1 / wave_shl:1); only the two edge lanes of a wave load them,
//   3. results leave with non-temporal 16-byte stores (`out` is never re-read).
// No LDS and no barriers: the J reuse lives in registers, the strip's two halo rows are L2 hits.
// Tuning data: profiles/r1_microbench_*.log (LJ, block size, nt loads/stores, XCD order).
#pragma once

#include "common.hip.h"
#include "lane_shift.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

// c = in[0,0], w = in[-1,0], e = in[+1,0], s = in[0,-1], n = in[0,+1].
// T: field dtype, W: dtype of the float literals' promotion (double unless literal precision 32
// on float fields) -- /root/reference/src/gt4py/cartesian/gtc/passes/gtir_upcaster.py:43-143.
template <typename T, typename W, int VARIANT>
__device__ __forceinline__ T lap5_expr(T c, T w, T e, T s, T n) {
    if constexpr (VARIANT == GT4MI_LAP_NOTEBOOK) {
        // ((((-4.0*c) + w) + e) + s) + n
        W r = (W)(-4.0) * (W)c;
        r = r + (W)w;
        r = r + (W)e;
        r = r + (W)s;
        r = r + (W)n;
        return (T)r;
    } else if constexpr (VARIANT == GT4MI_LAP_DOCS) {
        // (-4.0*c) + (((e + w) + n) + s)   -- the bracket is evaluated in the field dtype
        T sum = ((e + w) + n) + s;
        return (T)(((W)(-4.0) * (W)c) + (W)sum);
    } else if constexpr (VARIANT == GT4MI_LAP_SUITE) {
        // (4.0*c) - (((e + w) + n) + s)
        T sum = ((e + w) + n) + s;
        return (T)(((W)(4.0) * (W)c) - (W)sum);
    } else {
        // 0.25 * (((n + s) + e) + w)
        T sum = ((n + s) + e) + w;
        return (T)((W)(0.25) * (W)sum);
    }
}

// The strip of ONE lane: VEC columns from i0 x rows [j0, min(j0+LJ, dJ)) of level k.  edge_w / edge_e: the lane's west / east
// neighbour column is not in the adjacent lane's registers (first / last lane of a row of lanes) and is loaded instead.
// MASKED: only the columns [c_lo, c_hi) of the (16-byte aligned) view are the compute domain -- a domain whose first column
// is not on a 16-byte boundary, or whose width is odd, shifted onto the boundary (lap5_launch_variant): vectors are loaded
// whole (they reach at most one column into the halo on either side), stored element by element where they straddle the
// domain's edge, and a neighbour column outside the halo is never loaded (its consumer is not a domain point).
template <typename T, typename W, int VARIANT, int VEC, int LJ, int NTL = 0, bool MASKED = false>
__device__ __forceinline__ void lap5_strip_lane(const View<const T>& in, const View<T>& out, int dJ, int i0, bool active,
                                                bool edge_w, bool edge_e, int j0, unsigned k, int c_lo = 0, int c_hi = 0) {
    const T* __restrict__ col = in.p + (int64_t)k * in.sk + i0;
    T* __restrict__ ocol = out.p + (int64_t)k * out.sk + i0;

    // Row t of the register tile is `in` row min(j0 - 1 + t, dJ): rows past the strip's last valid
    // halo row are clamped (their results are never stored).
    T r[LJ + 2][VEC];
    int64_t roff[LJ + 2];
#pragma unroll
    for (int t = 0; t < LJ + 2; ++t) {
        int jr = j0 - 1 + t;
        jr = jr > dJ ? dJ : jr;
        roff[t] = (int64_t)jr * in.sj;
        // NTL: rows no other strip needs (t = 2 .. LJ-1) may be loaded non-temporally so that they do not
        // displace the rows neighbouring strips share in L2 (1 = those rows only, 2 = every row)
        if constexpr (NTL == 2 || (NTL == 1 && VEC == 2 && sizeof(T) == 8)) {
            if (NTL == 2 || (t >= 2 && t <= LJ - 1)) {
                using V = typename VecT<T, VEC>::type;
                const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(col + roff[t]));
#pragma unroll
                for (int q = 0; q < VEC; ++q) r[t][q] = v[q];
            } else {
                vload<T, VEC>(col + roff[t], r[t]);
            }
        } else {
            vload<T, VEC>(col + roff[t], r[t]);
        }
    }
    bool first_in = true, last_in = true;  // the lane's first / last column is a domain point
    if constexpr (MASKED) {
        first_in = i0 >= c_lo && i0 < c_hi;
        last_in = i0 + VEC - 1 >= c_lo && i0 + VEC - 1 < c_hi;
    }
    T w[LJ + 2], e[LJ + 2];
#pragma unroll
    for (int t = 1; t <= LJ; ++t) {
        T wl = lane_shift<T, true>(r[t][VEC - 1]);
        T el = lane_shift<T, false>(r[t][0]);
        if (edge_w && first_in) wl = col[roff[t] - 1];
        if (edge_e && last_in) el = col[roff[t] + VEC];
        w[t] = wl;
        e[t] = el;
    }
#pragma unroll
    for (int t = 1; t <= LJ; ++t) {
        T res[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const T wv = (v == 0) ? w[t] : r[t][v - 1];
            const T ev = (v == VEC - 1) ? e[t] : r[t][v + 1];
            res[v] = lap5_expr<T, W, VARIANT>(r[t][v], wv, ev, r[t - 1][v], r[t + 1][v]);
        }
        if (active && (j0 + t - 1 < dJ)) {
            T* o = ocol + (int64_t)(j0 + t - 1) * out.sj;
            if (!MASKED || (first_in && last_in)) {
                vstore<T, VEC, true>(o, res);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (i0 + v >= c_lo && i0 + v < c_hi) __builtin_nontemporal_store(res[v], o + v);
            }
        }
    }
}

// One strip: columns [bx*BLOCK*VEC, ...) x rows [j0, min(j0+LJ, dJ)) of level k; the lanes of a wave lie along I.
template <typename T, typename W, int VARIANT, int VEC, int LJ, int BLOCK, int NTL = 0, bool MASKED = false>
__device__ __forceinline__ void lap5_strip_tile(const View<const T>& in, const View<T>& out, int dI, int dJ,
                                                unsigned bx, int j0, unsigned k, int c_lo = 0) {
    const unsigned lane = threadIdx.x & 63;
    // Lanes past the end of the row stay active (DPP needs their neighbours' exec bits) but are
    // clamped onto the last valid vector and never store.  (MASKED: dI counts from the aligned column, the domain is
    // [c_lo, dI); the last vector may hold one column of halo.)
    int i0 = (int)(bx * BLOCK + threadIdx.x) * VEC;
    const bool active = i0 < dI;
    if (!active) i0 = MASKED ? ((dI + VEC - 1) / VEC - 1) * VEC : dI - VEC;
    lap5_strip_lane<T, W, VARIANT, VEC, LJ, NTL, MASKED>(in, out, dJ, i0, active, lane == 0, (lane == 63) || (i0 + VEC >= dI), j0, k,
                                                        c_lo, dI);
}

// A box only LPR * VEC columns wide (the W / E boxes of a decomposed apply): a wave is 64 / LPR rows of LPR lanes, each row of
// lanes a strip of LJ rows of its own -- the lane shifts of lap5_strip_lane still fetch the neighbour column inside a row of
// lanes, the first and last lane of each row load theirs.  A wave covers (64 / LPR) * LJ rows of the box.
template <typename T, typename W, int VARIANT, int VEC, int LJ, int LPR>
__device__ __forceinline__ void lap5_narrow_tile(const View<const T>& in, const View<T>& out, int dI, int dJ, unsigned tile,
                                                 unsigned k) {
    const int lane = (int)(threadIdx.x & 63);
    const int li = lane % LPR, sub = lane / LPR;
    int i0 = li * VEC;
    const bool active = i0 < dI;
    if (!active) i0 = dI - VEC;
    const int j0 = ((int)tile * (64 / LPR) + sub) * LJ;
    lap5_strip_lane<T, W, VARIANT, VEC, LJ>(in, out, dJ, i0, active && j0 < dJ, li == 0, (li == LPR - 1) || (i0 + VEC >= dI),
                                            j0 < dJ ? j0 : dJ, k);
}

template <typename T, typename W, int VARIANT, int VEC, int LJ, int BLOCK, int XCDG = 0, int NTL = 0, bool MASKED = false>
__global__ void __launch_bounds__(BLOCK)
lap5_strip_kernel(View<const T> in, View<T> out, int dI, int dJ, unsigned tiles_x, unsigned tiles_y, int c_lo = 0) {
    // XCDG > 0: runs of XCDG consecutive strips share an XCD (private L2), so the halo rows they
    // share are L2 hits instead of a second fabric fetch.  XCDG = -1: one contiguous range per XCD.
    unsigned b = blockIdx.x;
    if constexpr (XCDG > 0) b = xcd_remap_grouped<(unsigned)XCDG>(b, gridDim.x);
    if constexpr (XCDG < 0) b = xcd_remap(b, gridDim.x);
    const unsigned bx = b % tiles_x;
    const unsigned by = (b / tiles_x) % tiles_y;
    const unsigned k = b / (tiles_x * tiles_y);
    lap5_strip_tile<T, W, VARIANT, VEC, LJ, BLOCK, NTL, MASKED>(in, out, dI, dJ, bx, (int)by * LJ, k, c_lo);
}

// Up to two single J rows (row_a, row_b) of every level in ONE launch: the boundary strips of a
// J-decomposed domain, which wait for the halo exchange (gt4mi_dist_lap5_f64).
template <typename T, typename W, int VARIANT, int VEC, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
lap5_rows_kernel(View<const T> in, View<T> out, int dI, int row_a, int row_b, unsigned tiles_x, unsigned nrows) {
    const unsigned b = blockIdx.x;
    const unsigned bx = b % tiles_x;
    const unsigned r = (b / tiles_x) % nrows;
    const unsigned k = b / (tiles_x * nrows);
    const int j = r == 0 ? row_a : row_b;
    lap5_strip_tile<T, W, VARIANT, VEC, 1, BLOCK>(in, out, dI, j + 1, bx, j, k);
}

// Any-stride fallback: one thread per point, I fastest across lanes.  Correct for every layout the
// reference accepts (stencil_object.py:412-425 only warns about non-optimal layouts).
template <typename T, typename W, int VARIANT>
__global__ void __launch_bounds__(256)
lap5_generic_kernel(View<const T> in, View<T> out, int dI, int dJ, int dK) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= dI || j >= dJ) return;
    for (int k = blockIdx.z; k < dK; k += gridDim.z) {
        const T* p = in.p + (int64_t)i * in.si + (int64_t)j * in.sj + (int64_t)k * in.sk;
        const T r = lap5_expr<T, W, VARIANT>(p[0], p[-in.si], p[in.si], p[-in.sj], p[in.sj]);
        out.p[(int64_t)i * out.si + (int64_t)j * out.sj + (int64_t)k * out.sk] = r;
    }
}

// ---- launch configuration ------------------------------------------------------------------
struct Lap5Tuning {
    static constexpr int LJ = 8;  // rows per strip; all LJ+2 row loads are in flight at once
    // Runs of 4 consecutive strips per XCD: measured 382 vs 372 GLUPS (none) vs 370 (one contiguous
    // range per XCD) at 512^3, and L2->fabric reads 1.13x instead of 1.33x the algorithmic bytes
    // (profiles/r1_microbench_g_xcd_groups.log, r1_lap512_fetch_by_variant.txt).
    static constexpr int XCDG = 4;
};

template <typename T, typename W, int VARIANT, int VEC, int BLOCK>
inline int lap5_launch_strip(const View<const T>& in, const View<T>& out, const int64_t d[3],
                             hipStream_t stream) {
    constexpr int LJ = Lap5Tuning::LJ;
    const unsigned tx = (unsigned)cdiv(d[0], (int64_t)BLOCK * VEC);
    const unsigned ty = (unsigned)cdiv(d[1], LJ);
    const int64_t n = (int64_t)tx * ty * d[2];
    if (n > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "lap5: domain too large for one launch");
    hipLaunchKernelGGL((lap5_strip_kernel<T, W, VARIANT, VEC, LJ, BLOCK, Lap5Tuning::XCDG>), dim3((unsigned)n), dim3(BLOCK),
                       launch_dynamic_lds(), stream, in, out, (int)d[0], (int)d[1], tx, ty, 0);
    return GT4MI_OK;
}

// The same strips for a domain that starts `lead` columns past a 16-byte boundary and / or has an odd width: the views are
// moved onto the boundary and the kernel masks the columns outside [lead, lead + dI) (lap5_strip_lane<..., MASKED>).
template <typename T, typename W, int VARIANT, int VEC, int BLOCK>
inline int lap5_launch_strip_masked(const View<const T>& in, const View<T>& out, const int64_t d[3], int lead, hipStream_t stream) {
    constexpr int LJ = Lap5Tuning::LJ;
    const View<const T> in_a{in.p - lead, in.si, in.sj, in.sk};
    const View<T> out_a{out.p - lead, out.si, out.sj, out.sk};
    const int64_t width = d[0] + lead;
    const unsigned tx = (unsigned)cdiv(width, (int64_t)BLOCK * VEC);
    const unsigned ty = (unsigned)cdiv(d[1], LJ);
    const int64_t n = (int64_t)tx * ty * d[2];
    if (n > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "lap5: domain too large for one launch");
    hipLaunchKernelGGL((lap5_strip_kernel<T, W, VARIANT, VEC, LJ, BLOCK, Lap5Tuning::XCDG, 0, true>), dim3((unsigned)n), dim3(BLOCK),
                       launch_dynamic_lds(), stream, in_a, out_a, (int)width, (int)d[1], tx, ty, lead);
    return GT4MI_OK;
}

template <typename T, typename W, int VARIANT>
inline int lap5_launch_variant(const View<const T>& in, const View<T>& out, const int64_t d[3],
                               hipStream_t stream) {
    constexpr int VMAX = 16 / sizeof(T);
    // (domains only a few columns wide -- west / east boundary strips -- go to the thread-per-point kernel)
    if (in.si == 1 && out.si == 1 && d[0] >= 16) {
        const bool vec = vec_ok(in, VMAX) && vec_ok(out, VMAX) && (d[0] % VMAX == 0);
        if (vec) {
            const int64_t lanes = d[0] / VMAX;
            if (lanes <= 64) return lap5_launch_strip<T, W, VARIANT, VMAX, 64>(in, out, d, stream);
            if (lanes <= 128) return lap5_launch_strip<T, W, VARIANT, VMAX, 128>(in, out, d, stream);
            return lap5_launch_strip<T, W, VARIANT, VMAX, 256>(in, out, d, stream);
        }
        {
            // Rows that are aligned among themselves to a lane of TWO items (16 bytes for 8-byte items, 8 bytes for 4-byte
            // ones: a whole vector then reaches at most one column into the halo on either side), `inp` and `out` equally far
            // (0 or 1 column) from such a boundary: two-item lanes with masked edges instead of one-item lanes (a domain origin
            // that is not the storage's aligned column, or an odd width, cost 14-19 % on 8-byte lanes:
            // profiles/r3_misaligned_origin.log)
            constexpr int VM = 2;
            constexpr uintptr_t unit = VM * sizeof(T);
            const int lead = (int)((reinterpret_cast<uintptr_t>(in.p) % unit) / sizeof(T));
            const bool same = lead == (int)((reinterpret_cast<uintptr_t>(out.p) % unit) / sizeof(T));
            if (same && in.sj % VM == 0 && in.sk % VM == 0 && out.sj % VM == 0 && out.sk % VM == 0) {
                const int64_t lanes = cdiv(d[0] + lead, VM);
                if (lanes <= 64) return lap5_launch_strip_masked<T, W, VARIANT, VM, 64>(in, out, d, lead, stream);
                if (lanes <= 128) return lap5_launch_strip_masked<T, W, VARIANT, VM, 128>(in, out, d, lead, stream);
                // one lane more than the aligned domain needs (512 columns from an odd origin: 257 lanes) must not cost a
                // second, almost empty workgroup per row: five waves instead of four where that wastes fewer lanes
                if (cdiv(lanes, 320) * 320 < cdiv(lanes, 256) * 256)
                    return lap5_launch_strip_masked<T, W, VARIANT, VM, 320>(in, out, d, lead, stream);
                return lap5_launch_strip_masked<T, W, VARIANT, VM, 256>(in, out, d, lead, stream);
            }
        }
        if (d[0] <= 64) return lap5_launch_strip<T, W, VARIANT, 1, 64>(in, out, d, stream);
        return lap5_launch_strip<T, W, VARIANT, 1, 256>(in, out, d, stream);
    }
    dim3 grid((unsigned)cdiv(d[0], 64), (unsigned)cdiv(d[1], 4),
              (unsigned)(d[2] < 65535 ? d[2] : 65535));
    hipLaunchKernelGGL((lap5_generic_kernel<T, W, VARIANT>), grid, dim3(256), 0, stream, in, out,
                       (int)d[0], (int)d[1], (int)d[2]);
    return GT4MI_OK;
}

template <typename T, typename W>
inline int lap5_run(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* outf,
                    int variant, hipStream_t stream) {
    if (int rc = check_domain(domain)) return rc;
    const int h1[3] = {1, 1, 0}, h0[3] = {0, 0, 0};
    View<T> in_v, out_v;
    if (int rc = make_view<T>("inp", inp, domain, h1, h1, &in_v)) return rc;
    if (int rc = make_view<T>("out", outf, domain, h0, h0, &out_v)) return rc;
    if (domain[0] == 0 || domain[1] == 0 || domain[2] == 0) return GT4MI_OK;
    if (views_overlap(in_v, h1, h1, out_v, h0, h0, domain))
        return fail(GT4MI_ERR_UNSUPPORTED,
                    "lap5: 'inp' and 'out' overlap in memory; every point reads its neighbours' OLD values (the "
                    "reference evaluates the right-hand side before it assigns), which an in-place kernel cannot "
                    "provide -- pass a separate output array");
    View<const T> in_c{in_v.p, in_v.si, in_v.sj, in_v.sk};
    int rc;
    switch (variant) {
        case GT4MI_LAP_NOTEBOOK: rc = lap5_launch_variant<T, W, GT4MI_LAP_NOTEBOOK>(in_c, out_v, domain, stream); break;
        case GT4MI_LAP_DOCS: rc = lap5_launch_variant<T, W, GT4MI_LAP_DOCS>(in_c, out_v, domain, stream); break;
        case GT4MI_LAP_SUITE: rc = lap5_launch_variant<T, W, GT4MI_LAP_SUITE>(in_c, out_v, domain, stream); break;
        case GT4MI_LAP_AVG: rc = lap5_launch_variant<T, W, GT4MI_LAP_AVG>(in_c, out_v, domain, stream); break;
        default: return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5: unknown variant %d", variant);
    }
    if (rc) return rc;
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

// Rows row_a (and row_b when nrows == 2) of the compute domain, all K levels, in one launch.
// Falls back to per-row lap5_run calls when the fields do not qualify for the vector path.
template <typename T, typename W, int VARIANT>
inline int lap5_rows_variant(const View<const T>& in, const View<T>& out, const int64_t d[3], int row_a, int row_b,
                             int nrows, hipStream_t stream, bool* done) {
    constexpr int VMAX = 16 / sizeof(T);
    *done = false;
    if (!(in.si == 1 && out.si == 1 && vec_ok(in, VMAX) && vec_ok(out, VMAX) && d[0] % VMAX == 0)) return GT4MI_OK;
    const unsigned tx = (unsigned)cdiv(d[0], (int64_t)256 * VMAX);
    hipLaunchKernelGGL((lap5_rows_kernel<T, W, VARIANT, VMAX, 256>), dim3(tx * (unsigned)nrows * (unsigned)d[2]), dim3(256), 0,
                       stream, in, out, (int)d[0], row_a, row_b, tx, (unsigned)nrows);
    *done = true;
    return GT4MI_OK;
}

template <typename T, typename W>
inline int lap5_run_rows(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* outf, int variant,
                         int row_a, int row_b, int nrows, hipStream_t stream) {
    if (nrows <= 0 || domain[0] <= 0 || domain[2] <= 0) return GT4MI_OK;
    if (int rc = check_domain(domain)) return rc;
    const int h1[3] = {1, 1, 0}, h0[3] = {0, 0, 0};
    View<T> in_v, out_v;
    if (int rc = make_view<T>("inp", inp, domain, h1, h1, &in_v)) return rc;
    if (int rc = make_view<T>("out", outf, domain, h0, h0, &out_v)) return rc;
    if (views_overlap(in_v, h1, h1, out_v, h0, h0, domain))
        return fail(GT4MI_ERR_UNSUPPORTED, "lap5: 'inp' and 'out' overlap in memory (see gt4mi_lap5_*)");
    View<const T> in_c{in_v.p, in_v.si, in_v.sj, in_v.sk};
    bool done = false;
    switch (variant) {
        case GT4MI_LAP_NOTEBOOK: lap5_rows_variant<T, W, GT4MI_LAP_NOTEBOOK>(in_c, out_v, domain, row_a, row_b, nrows, stream, &done); break;
        case GT4MI_LAP_DOCS: lap5_rows_variant<T, W, GT4MI_LAP_DOCS>(in_c, out_v, domain, row_a, row_b, nrows, stream, &done); break;
        case GT4MI_LAP_SUITE: lap5_rows_variant<T, W, GT4MI_LAP_SUITE>(in_c, out_v, domain, row_a, row_b, nrows, stream, &done); break;
        case GT4MI_LAP_AVG: lap5_rows_variant<T, W, GT4MI_LAP_AVG>(in_c, out_v, domain, row_a, row_b, nrows, stream, &done); break;
        default: return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5: unknown variant %d", variant);
    }
    if (done) {
        GT4MI_HIP_CHECK(hipGetLastError());
        return GT4MI_OK;
    }
    for (int r = 0; r < nrows; ++r) {  // generic layouts: one ordinary launch per row
        gt4mi_field a = *inp, b = *outf;
        const int j = r == 0 ? row_a : row_b;
        a.origin[1] += j;
        b.origin[1] += j;
        const int64_t d1[3] = {domain[0], 1, domain[2]};
        if (int rc = lap5_run<T, W>(d1, &a, &b, variant, stream)) return rc;
    }
    return GT4MI_OK;
}

}  // namespace gt4mi
