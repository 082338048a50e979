// Horizontal diffusion on the BOUNDARY RING of a domain in one launch.
//
// An IJ-decomposed apply computes the interior of its local domain while the ghost cells travel and
// the ring of points that read them -- as deep as the stencil's halo: 2 -- afterwards.  The ring is four boxes:
//
//      +-------------------------------+      S / N: `lo_j` / `hi_j` rows over the whole width  (row boxes)
//      |               N               |      W / E: `lo_i` / `hi_i` columns between them       (column boxes)
//      +---+-----------------------+---+
//      | W |       interior        | E |
//      +---+-----------------------+---+
//      |               S               |
//      +-------------------------------+
//
// Four launches of the whole-domain kernels cost more than the work: the row boxes are fine for the J-march kernel
// (hdiff_jmarch.hip.h) but in a 2-column box 62 of its 64 lanes idle, and the thread-per-point fallback took 29 us for
// the two column boxes of a 512 x 1024 x 80 share next to 182 us for the interior (profiles/r1_dist_hdiff_rehearsal.log).
// Here ONE launch covers all four boxes; every wave of a workgroup takes one tile of one box:
//   row boxes     the J-march strip (lanes along I, 16-byte vectors, DPP shifts along I), strips of 2 rows;
//   column boxes  the same design TRANSPOSED: lanes along J (one row each), NC = 2 output columns per lane from the
//                 NC + 4 columns of `in` the lane loads itself; the J neighbours of `in`, `lap` and `fly` come from the
//                 adjacent lanes with DPP wave shifts.  Two halo lanes at either end of a wave only feed their
//                 neighbours, consecutive waves overlap by four rows.
// Same per-point arithmetic (hd_lap / hd_flux / hd_out) as every other kernel of the family: bit-identical results.
#pragma once

#include "hdiff.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

struct RingBoxes {
    static constexpr int MAX = 4;
    int n;                      // boxes in use
    int kind[MAX];              // 0 = row box (strips of 2 rows), 1 = 2-column box (lanes along J), 2 = tall box (J-march strips)
    int i0[MAX], j0[MAX];       // first point, relative to the compute-domain origin
    int ei[MAX], ej[MAX];       // extent
    unsigned per_level[MAX];    // tiles per K level
    unsigned tiles_i[MAX];      // row boxes: tiles along I
    int lead[MAX];              // J-march boxes: items the box's first column lies past a 16-byte boundary (hdiff_jmarch_strip)
    unsigned first[MAX + 1];    // prefix sums of tiles (per_level * dK)
};

// Column box: NC columns x ej rows; lane l of tile t owns row t * 60 - 2 + l.
template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int NC>
__device__ __forceinline__ void hdiff_column_strip(const View<const T>& in, const View<T>& out, const View<const T>& cf,
                                                   PW coeff_scalar, int dJ, unsigned tile, unsigned k) {
    constexpr int HL = 2;  // halo lanes per side: fly[j-1] needs lap[j-1], which needs in[j-2]
    const int lane = (int)(threadIdx.x & 63);
    const int j = (int)tile * (64 - 2 * HL) - HL + lane;
    const bool readable = j >= -2 && j < dJ + 2;
    const bool writes = lane >= HL && lane < 64 - HL && j < dJ;
    const T* __restrict__ ip = in.p + (int64_t)k * in.sk + (int64_t)j * in.sj;
    T a[NC + 4];  // in[-2 .. NC + 2) of this lane's row
#pragma unroll
    for (int c = 0; c < NC + 4; ++c) a[c] = readable ? ip[c - 2] : (T)0;
    T cfr[NC];
    if constexpr (COEFF_FIELD) {
        const T* __restrict__ cp = cf.p + (int64_t)k * cf.sk + (int64_t)j * cf.sj;
#pragma unroll
        for (int c = 0; c < NC; ++c) cfr[c] = writes ? cp[c] : (T)0;
    }
    // lap on columns [-1, NC + 1) of this row; the rows above / below are the neighbouring lanes' `a`
    T up[NC + 2];  // in[c, j + 1] for c in [-1, NC + 1)
    W lap[NC + 2];
#pragma unroll
    for (int c = 0; c < NC + 2; ++c) {
        const T dn = lane_shift<T, true>(a[c + 1]);  // row j - 1
        up[c] = lane_shift<T, false>(a[c + 1]);      // row j + 1
        lap[c] = hd_lap<T, W>(a[c + 1], a[c + 2], a[c], up[c], dn);
    }
    W flx[NC + 1];  // columns [-1, NC)
#pragma unroll
    for (int c = 0; c < NC + 1; ++c) flx[c] = hd_flux<T, W, LIMITER>(lap[c + 1], lap[c], a[c + 2], a[c + 1]);
    T res[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const W lap_up = lane_shift<W, false>(lap[c + 1]);
        const W fly = hd_flux<T, W, LIMITER>(lap_up, lap[c + 1], up[c + 1], a[c + 2]);
        const W fly_dn = lane_shift<W, true>(fly);
        PW cv;
        if constexpr (COEFF_FIELD) cv = (PW)cfr[c];
        else cv = coeff_scalar;
        res[c] = hd_out<T, W, PW>(a[c + 2], cv, flx[c + 1], flx[c], fly, fly_dn);
    }
    if (writes) {
        T* __restrict__ op = out.p + (int64_t)k * out.sk + (int64_t)j * out.sj;
#pragma unroll
        for (int c = 0; c < NC; ++c) op[c] = res[c];
    }
}

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, int VEC>
__global__ void __launch_bounds__(256)
hdiff_ring_kernel(View<const T> in, View<T> out, View<const T> cf, PW coeff_scalar, RingBoxes b) {
    // wave-uniform, and the compiler is told so: the box table is then read with scalar loads
    const unsigned tile = blockIdx.x * 4 + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile >= b.first[b.n]) return;
    int m = 0;
    while (tile >= b.first[m + 1]) ++m;
    const unsigned t = tile - b.first[m];
    const unsigned k = t / b.per_level[m], r = t % b.per_level[m];
    const int64_t oi = b.i0[m], oj = b.j0[m];
    const View<const T> in_b{in.p + oi + oj * in.sj, 1, in.sj, in.sk};
    const View<T> out_b{out.p + oi + oj * out.sj, 1, out.sj, out.sk};
    const View<const T> cf_b{COEFF_FIELD ? cf.p + oi + oj * cf.sj : nullptr, 1, cf.sj, cf.sk};
    if (b.kind[m] == 0) {
        hdiff_jmarch_strip<T, W, PW, LIMITER, COEFF_FIELD, VEC, 2, 2>(in_b, out_b, cf_b, coeff_scalar, b.ei[m], b.ej[m],
                                                                      r % b.tiles_i[m], r / b.tiles_i[m], k, b.lead[m]);
    } else if (b.kind[m] == 2) {
        // a W / E box several columns wide: the whole-domain kernel's strips (full cache lines, nothing strided)
        hdiff_jmarch_strip<T, W, PW, LIMITER, COEFF_FIELD, VEC, HdiffTuning<T>::LJ, HdiffTuning<T>::PF>(
            in_b, out_b, cf_b, coeff_scalar, b.ei[m], b.ej[m], r % b.tiles_i[m], r / b.tiles_i[m], k, b.lead[m]);
    } else {
        hdiff_column_strip<T, W, PW, LIMITER, COEFF_FIELD, 2>(in_b, out_b, cf_b, coeff_scalar, b.ej[m], r, k);
    }
}

// Widths of the ring towards low I, high I, low J, high J (0 where the domain has no neighbour).  The ring kernel takes
// I-contiguous fields; a W / E box exactly 2 columns wide runs the transposed tile, any other width J-march strips --
// the distributed step asks for W / E boxes a few cache lines wide (GT4MI_PLAN_EDGE_COLUMNS): what they compute is taken out
// of the interior kernel, and a box of 32 columns moves whole 256-byte row segments where the 2-column box touched four
// sectors per row for 16 bytes of output.  Other layouts run box by box on the ordinary kernels.
template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD>
inline int hdiff_launch_ring(const View<const T>& in, const View<T>& out, const View<const T>& cf, PW coeff_scalar,
                             const int64_t d[3], const int widths[4], hipStream_t stream, bool point_per_thread = false) {
    const int64_t di = d[0], dj = d[1], dk = d[2];
    const int64_t lo_i = widths[0], hi_i = widths[1], lo_j = widths[2], hi_j = widths[3];
    struct Box { int kind; int64_t i0, j0, ei, ej; } boxes[4];
    int n = 0;
    if (lo_j > 0) boxes[n++] = {0, 0, 0, di, lo_j};
    if (hi_j > 0) boxes[n++] = {0, 0, dj - hi_j, di, hi_j};
    if (lo_i > 0 && dj - lo_j - hi_j > 0) boxes[n++] = {1, 0, lo_j, lo_i, dj - lo_j - hi_j};
    if (hi_i > 0 && dj - lo_j - hi_j > 0) boxes[n++] = {1, di - hi_i, lo_j, hi_i, dj - lo_j - hi_j};
    if (n == 0 || dk == 0) return GT4MI_OK;
    const bool contiguous = in.si == 1 && out.si == 1 && (!COEFF_FIELD || cf.si == 1);
    const bool fits = contiguous && !point_per_thread && hdiff_jmarch_enabled();  // point_per_thread: `coeff` IS `out_field` (hdiff.hip.h)
    for (int m = 0; m < n; ++m)
        if (boxes[m].kind == 1 && boxes[m].ei != 2) boxes[m].kind = 2;  // not the 2-column shape: J-march strips
    if (!fits) {
        for (int m = 0; m < n; ++m) {
            const Box& x = boxes[m];
            const int64_t sub[3] = {x.ei, x.ej, dk};
            const View<const T> in_b{in.p + x.i0 * in.si + x.j0 * in.sj, in.si, in.sj, in.sk};
            const View<T> out_b{out.p + x.i0 * out.si + x.j0 * out.sj, out.si, out.sj, out.sk};
            const View<const T> cf_b{COEFF_FIELD ? cf.p + x.i0 * cf.si + x.j0 * cf.sj : nullptr, cf.si, cf.sj, cf.sk};
            if (int rc = hdiff_launch<T, W, PW, LIMITER, COEFF_FIELD>(in_b, out_b, cf_b, coeff_scalar, sub, stream, point_per_thread))
                return rc;
        }
        return GT4MI_OK;
    }
    constexpr int VMAX = 16 / sizeof(T);
    int base_lead = 0;
    const bool vec = hdiff_common_lead<T, COEFF_FIELD>(in, out, cf, VMAX, &base_lead);
    RingBoxes b;
    b.n = n;
    b.first[0] = 0;
    for (int m = 0; m < RingBoxes::MAX; ++m) {
        if (m >= n) {
            b.kind[m] = b.i0[m] = b.j0[m] = b.ei[m] = b.ej[m] = b.lead[m] = 0;
            b.per_level[m] = b.tiles_i[m] = 1;
            b.first[m + 1] = b.first[n];
            continue;
        }
        const Box& x = boxes[m];
        b.kind[m] = x.kind; b.i0[m] = (int)x.i0; b.j0[m] = (int)x.j0; b.ei[m] = (int)x.ei; b.ej[m] = (int)x.ej;
        b.lead[m] = vec ? (int)(((base_lead + x.i0) % VMAX + VMAX) % VMAX) : 0;  // of the box's own first column
        if (x.kind != 1) {
            const int out_lanes = vec ? 62 * VMAX : 60;  // columns per wave (hdiff_jmarch_strip: H = 1 for vectors, 2 for scalars)
            b.tiles_i[m] = (unsigned)cdiv(x.ei + b.lead[m], out_lanes);
            b.per_level[m] = b.tiles_i[m] * (unsigned)cdiv(x.ej, x.kind == 0 ? 2 : HdiffTuning<T>::LJ);
        } else {
            b.tiles_i[m] = 1;
            b.per_level[m] = (unsigned)cdiv(x.ej, 60);
        }
        const int64_t total = (int64_t)b.first[m] + (int64_t)b.per_level[m] * dk;
        if (total > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "hdiff ring: domain too large for one launch");
        b.first[m + 1] = (unsigned)total;
    }
    const unsigned blocks = (unsigned)cdiv(b.first[n], 4);
    if (vec)
        hipLaunchKernelGGL((hdiff_ring_kernel<T, W, PW, LIMITER, COEFF_FIELD, VMAX>), dim3(blocks), dim3(256), 0, stream, in, out, cf,
                           coeff_scalar, b);
    else
        hipLaunchKernelGGL((hdiff_ring_kernel<T, W, PW, LIMITER, COEFF_FIELD, 1>), dim3(blocks), dim3(256), 0, stream, in, out, cf,
                           coeff_scalar, b);
    return GT4MI_OK;
}

template <typename T, typename W, typename PW>
inline int hdiff_ring_dispatch(const View<const T>& in, const View<T>& out, const View<const T>& cf, bool coeff_field,
                               PW coeff_scalar, bool limiter, const int64_t d[3], const int widths[4], hipStream_t stream,
                               bool point_per_thread) {
    if (limiter) {
        if (coeff_field) return hdiff_launch_ring<T, W, PW, true, true>(in, out, cf, coeff_scalar, d, widths, stream, point_per_thread);
        return hdiff_launch_ring<T, W, PW, true, false>(in, out, cf, coeff_scalar, d, widths, stream, point_per_thread);
    }
    if (coeff_field) return hdiff_launch_ring<T, W, PW, false, true>(in, out, cf, coeff_scalar, d, widths, stream, point_per_thread);
    return hdiff_launch_ring<T, W, PW, false, false>(in, out, cf, coeff_scalar, d, widths, stream, point_per_thread);
}

// The ring of `domain` (widths towards W, E, S, N; each 0 or <= the domain) -- same argument rules as hdiff_run.
template <typename T>
inline int hdiff_ring_run(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                          const gt4mi_field* coeff, double coeff_scalar, int flags, const int widths[4], hipStream_t stream) {
    if (int rc = check_domain(domain)) return rc;
    if (widths == nullptr) return fail(GT4MI_ERR_INVALID_ARGUMENT, "hdiff ring: widths is null");
    if (widths[0] < 0 || widths[1] < 0 || widths[2] < 0 || widths[3] < 0 || widths[0] + widths[1] > domain[0] ||
        widths[2] + widths[3] > domain[1])
        return fail(GT4MI_ERR_INVALID_ARGUMENT, "hdiff ring: widths (%d, %d, %d, %d) do not fit the %lld x %lld domain", widths[0],
                    widths[1], widths[2], widths[3], (long long)domain[0], (long long)domain[1]);
    const int h2[3] = {2, 2, 0}, h0[3] = {0, 0, 0};
    View<T> in_v, out_v, cf_v{nullptr, 0, 0, 0};
    if (int rc = make_view<T>("in_field", in_field, domain, h2, h2, &in_v)) return rc;
    if (int rc = make_view<T>("out_field", out_field, domain, h0, h0, &out_v)) return rc;
    if (coeff != nullptr)
        if (int rc = make_view<T>("coeff", coeff, domain, h0, h0, &cf_v)) return rc;
    if (domain[0] == 0 || domain[1] == 0 || domain[2] == 0) return GT4MI_OK;
    if (views_overlap(in_v, h2, h2, out_v, h0, h0, domain))
        return fail(GT4MI_ERR_UNSUPPORTED, "hdiff: 'in_field' and 'out_field' overlap in memory (see gt4mi_hdiff_*)");
    bool alias = false;  // out_field IS coeff (see hdiff_run)
    if (coeff != nullptr && views_overlap(cf_v, h0, h0, out_v, h0, h0, domain)) {
        if (!same_view(cf_v, out_v))
            return fail(GT4MI_ERR_UNSUPPORTED, "hdiff: 'coeff' and 'out_field' overlap in memory without being the same elements");
        alias = true;
    }
    const View<const T> in_c{in_v.p, in_v.si, in_v.sj, in_v.sk};
    const View<const T> cf_c{cf_v.p, cf_v.si, cf_v.sj, cf_v.sk};
    const bool limiter = (flags & GT4MI_HDIFF_LIMITER) != 0;
    const bool has_field = coeff != nullptr;
    int rc;
    if constexpr (sizeof(T) == 8) {
        const double cs = (flags & GT4MI_HDIFF_COEFF_F32) ? (double)(float)coeff_scalar : coeff_scalar;
        rc = hdiff_ring_dispatch<T, double, double>(in_c, out_v, cf_c, has_field, cs, limiter, domain, widths, stream, alias);
    } else {
        const bool w32 = (flags & GT4MI_HDIFF_INTERNAL_F32) != 0;
        const bool c32 = (flags & GT4MI_HDIFF_COEFF_F32) != 0;
        if (!w32) {
            const double cs = c32 ? (double)(float)coeff_scalar : coeff_scalar;
            rc = hdiff_ring_dispatch<T, double, double>(in_c, out_v, cf_c, has_field, cs, limiter, domain, widths, stream, alias);
        } else if (has_field || c32) {
            rc = hdiff_ring_dispatch<T, float, float>(in_c, out_v, cf_c, has_field, (float)coeff_scalar, limiter, domain, widths, stream, alias);
        } else {
            rc = hdiff_ring_dispatch<T, float, double>(in_c, out_v, cf_c, has_field, coeff_scalar, limiter, domain, widths, stream, alias);
        }
    }
    if (rc) return rc;
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

}  // namespace gt4mi
