// 5-point stencils on a RING of boxes around (and beyond) the boundary of a domain, in one launch.
//
// What an IJ-decomposed run computes apart from its interior:
//   * exchange-every-apply: the one-point-deep ring that reads the ghost cells, after the exchange has been joined;
//   * communication-avoiding time stepping with ghost regions H deep (gt4mi_dist_lap5_f64_skewed): for step s of a cycle
//     the band from H - s points OUTSIDE the domain to 2H - s points inside it, computed before the interior so that the
//     faces can travel while H interior kernels run.
// Both are "the domain grown by outer[side], minus the domain shrunk by inner[side]": up to four boxes -- two row boxes
// over the full width (S, N) and two column boxes between them (W, E).  Row boxes run the register-strip tile of
// lap5.hip.h (lanes along I); column boxes of 2 .. 16 columns run rows of 16 / VEC lanes (lap5_narrow_tile: the distributed apply
// asks for W / E boxes 16 columns wide, which keeps its interior kernel on 16-byte lanes), other widths a thread per row.
// One launch instead of four; same per-point expression (lap5_expr): bit-identical to the whole-domain kernel.
#pragma once

#include "hdiff_ring.hip.h"  // RingBoxes
#include "lap5.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

template <typename T, typename W, int VARIANT, int VEC, int LJ>
__global__ void __launch_bounds__(256)
lap5_ring_kernel(View<const T> in, View<T> out, RingBoxes b) {
    const unsigned blk = blockIdx.x;
    int m = 0;
    while (blk >= b.first[m + 1]) ++m;
    const unsigned t = blk - b.first[m];
    const unsigned k = t / b.per_level[m], r = t % b.per_level[m];
    const int64_t oi = b.i0[m], oj = b.j0[m];
    const View<const T> in_b{in.p + oi + oj * in.sj, 1, in.sj, in.sk};
    const View<T> out_b{out.p + oi + oj * out.sj, 1, out.sj, out.sk};
    if (b.kind[m] == 0) {
        lap5_strip_tile<T, W, VARIANT, VEC, LJ, 256>(in_b, out_b, b.ei[m], b.ej[m], r % b.tiles_i[m], (int)(r / b.tiles_i[m]) * LJ, k);
    } else if (b.kind[m] == 2) {
        // a W / E box up to 16 columns wide: every wave takes (64 / LPR) strips of 8 rows (lap5_narrow_tile)
        constexpr int LPR = 16 / VEC;
        lap5_narrow_tile<T, W, VARIANT, VEC, 8, LPR>(in_b, out_b, b.ei[m], b.ej[m], r * 4 + (threadIdx.x >> 6), k);
    } else {
        const int j = (int)(r * 256 + threadIdx.x);
        if (j >= b.ej[m]) return;
        const T* __restrict__ p = in_b.p + (int64_t)k * in_b.sk + (int64_t)j * in_b.sj;
        T* __restrict__ o = out_b.p + (int64_t)k * out_b.sk + (int64_t)j * out_b.sj;
        const int64_t sj = in_b.sj;
        T w = p[-1], c = p[0];
        for (int i = 0; i < b.ei[m]; ++i) {
            const T e = p[i + 1];
            o[i] = lap5_expr<T, W, VARIANT>(c, w, e, p[i - sj], p[i + sj]);
            w = c;
            c = e;
        }
    }
}

template <typename T, typename W, int VARIANT>
inline int lap5_launch_ring(const View<const T>& in, const View<T>& out, const int64_t d[3], const int outer[4],
                            const int inner[4], hipStream_t stream) {
    const int64_t di = d[0], dj = d[1], dk = d[2];
    const int64_t ow = outer[0], oe = outer[1], os = outer[2], on = outer[3];
    const int64_t iw = inner[0], ie = inner[1], is = inner[2], in_ = inner[3];
    struct Box { int kind; int64_t i0, j0, ei, ej; } boxes[4];
    int n = 0;
    if (os + is > 0) boxes[n++] = {0, -ow, -os, di + ow + oe, os + is};
    if (on + in_ > 0) boxes[n++] = {0, -ow, dj - in_, di + ow + oe, on + in_};
    if (ow + iw > 0 && dj - is - in_ > 0) boxes[n++] = {1, -ow, is, ow + iw, dj - is - in_};
    if (oe + ie > 0 && dj - is - in_ > 0) boxes[n++] = {1, di - ie, is, oe + ie, dj - is - in_};
    if (n == 0 || dk == 0) return GT4MI_OK;
    if (!(in.si == 1 && out.si == 1)) {  // other layouts: box by box on the any-stride kernel
        for (int m = 0; m < n; ++m) {
            const Box& x = boxes[m];
            const int64_t sub[3] = {x.ei, x.ej, dk};
            const View<const T> in_b{in.p + x.i0 * in.si + x.j0 * in.sj, in.si, in.sj, in.sk};
            const View<T> out_b{out.p + x.i0 * out.si + x.j0 * out.sj, out.si, out.sj, out.sk};
            if (int rc = lap5_launch_variant<T, W, VARIANT>(in_b, out_b, sub, stream)) return rc;
        }
        return GT4MI_OK;
    }
    constexpr int VMAX = 16 / sizeof(T);
    bool vec = true;
    int64_t deepest = 1;
    for (int m = 0; m < n; ++m)  // column boxes of 2 .. 16 columns: rows of lanes instead of a thread per row (lap5_narrow_tile)
        if (boxes[m].kind == 1 && boxes[m].ei >= 2 && boxes[m].ei <= 16) boxes[m].kind = 2;
    for (int m = 0; m < n; ++m) {
        const Box& x = boxes[m];
        if (x.kind == 1) continue;
        const View<const T> in_b{in.p + x.i0 + x.j0 * in.sj, 1, in.sj, in.sk};
        const View<T> out_b{out.p + x.i0 + x.j0 * out.sj, 1, out.sj, out.sk};
        vec = vec && vec_ok(in_b, VMAX) && vec_ok(out_b, VMAX) && x.ei % VMAX == 0;
        if (x.kind == 0) deepest = x.ej > deepest ? x.ej : deepest;
    }
    const int lj = deepest <= 1 ? 1 : (deepest <= 2 ? 2 : 4);
    const int vecw = vec ? VMAX : 1;
    RingBoxes b;
    b.n = n;
    b.first[0] = 0;
    for (int m = 0; m < RingBoxes::MAX; ++m) {
        if (m >= n) {
            b.kind[m] = b.i0[m] = b.j0[m] = b.ei[m] = b.ej[m] = 0;
            b.per_level[m] = b.tiles_i[m] = 1;
            b.first[m + 1] = 0xffffffffu;  // never selected
            continue;
        }
        const Box& x = boxes[m];
        b.kind[m] = x.kind; b.i0[m] = (int)x.i0; b.j0[m] = (int)x.j0; b.ei[m] = (int)x.ei; b.ej[m] = (int)x.ej;
        if (x.kind == 0) {
            b.tiles_i[m] = (unsigned)cdiv(x.ei, (int64_t)256 * vecw);
            b.per_level[m] = b.tiles_i[m] * (unsigned)cdiv(x.ej, lj);
        } else if (x.kind == 2) {
            const int64_t rows_per_wave = (64 / (16 / vecw)) * 8;  // (64 / LPR) strips of 8 rows
            b.tiles_i[m] = 1;
            b.per_level[m] = (unsigned)cdiv(cdiv(x.ej, rows_per_wave), 4);  // four waves per workgroup
        } else {
            b.tiles_i[m] = 1;
            b.per_level[m] = (unsigned)cdiv(x.ej, 256);
        }
        const int64_t total = (int64_t)b.first[m] + (int64_t)b.per_level[m] * dk;
        if (total > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "lap5 ring: domain too large for one launch");
        b.first[m + 1] = (unsigned)total;
    }
    const unsigned blocks = b.first[n];
#define GT4MI_LAP5_RING(V, L) \
    hipLaunchKernelGGL((lap5_ring_kernel<T, W, VARIANT, V, L>), dim3(blocks), dim3(256), 0, stream, in, out, b)
    if (vec) {
        if (lj == 1) GT4MI_LAP5_RING(VMAX, 1);
        else if (lj == 2) GT4MI_LAP5_RING(VMAX, 2);
        else GT4MI_LAP5_RING(VMAX, 4);
    } else {
        if (lj == 1) GT4MI_LAP5_RING(1, 1);
        else if (lj == 2) GT4MI_LAP5_RING(1, 2);
        else GT4MI_LAP5_RING(1, 4);
    }
#undef GT4MI_LAP5_RING
    return GT4MI_OK;
}

// The region (domain grown by outer[W, E, S, N]) minus (domain shrunk by inner[W, E, S, N]).  `inp` must be readable one
// point beyond the grown domain; `inp` and `out` must not overlap (as for gt4mi_lap5_*).
template <typename T, typename W>
inline int lap5_ring_run(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* outf, int variant,
                         const int outer[4], const int inner[4], hipStream_t stream) {
    if (int rc = check_domain(domain)) return rc;
    if (outer == nullptr || inner == nullptr) return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5 ring: null widths");
    for (int s = 0; s < 4; ++s)
        if (outer[s] < 0 || inner[s] < 0) return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5 ring: negative width");
    if (inner[0] + inner[1] > domain[0] || inner[2] + inner[3] > domain[1])
        return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5 ring: inner widths do not fit the %lld x %lld domain", (long long)domain[0],
                    (long long)domain[1]);
    if (domain[0] == 0 || domain[1] == 0 || domain[2] == 0) return GT4MI_OK;
    // bounds and aliasing on the grown domain
    gt4mi_field a = *inp, o = *outf;
    a.origin[0] -= outer[0]; a.origin[1] -= outer[2];
    o.origin[0] -= outer[0]; o.origin[1] -= outer[2];
    const int64_t grown[3] = {domain[0] + outer[0] + outer[1], domain[1] + outer[2] + outer[3], domain[2]};
    const int h1[3] = {1, 1, 0}, h0[3] = {0, 0, 0};
    View<T> in_g, out_g;
    if (int rc = make_view<T>("inp", &a, grown, h1, h1, &in_g)) return rc;
    if (int rc = make_view<T>("out", &o, grown, h0, h0, &out_g)) return rc;
    if (views_overlap(in_g, h1, h1, out_g, h0, h0, grown))
        return fail(GT4MI_ERR_UNSUPPORTED, "lap5: 'inp' and 'out' overlap in memory (see gt4mi_lap5_*)");
    View<T> in_v, out_v;
    if (int rc = make_view<T>("inp", inp, domain, h0, h0, &in_v)) return rc;
    if (int rc = make_view<T>("out", outf, domain, h0, h0, &out_v)) return rc;
    const View<const T> in_c{in_v.p, in_v.si, in_v.sj, in_v.sk};
    int rc;
    switch (variant) {
        case GT4MI_LAP_NOTEBOOK: rc = lap5_launch_ring<T, W, GT4MI_LAP_NOTEBOOK>(in_c, out_v, domain, outer, inner, stream); break;
        case GT4MI_LAP_DOCS: rc = lap5_launch_ring<T, W, GT4MI_LAP_DOCS>(in_c, out_v, domain, outer, inner, stream); break;
        case GT4MI_LAP_SUITE: rc = lap5_launch_ring<T, W, GT4MI_LAP_SUITE>(in_c, out_v, domain, outer, inner, stream); break;
        case GT4MI_LAP_AVG: rc = lap5_launch_ring<T, W, GT4MI_LAP_AVG>(in_c, out_v, domain, outer, inner, stream); break;
        default: return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5: unknown variant %d", variant);
    }
    if (rc) return rc;
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

}  // namespace gt4mi
