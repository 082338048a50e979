// The interior kernel of a distributed 5-point step and the PUSH of the direct transport in ONE launch.
//
// On the one-stream ("inline") schedule the pack kernel of the direct transport runs in front of the interior kernel: 8-9 us for
// two 262 KB faces (two dependent round trips and a launch, on an idle device) before the 46 us that matter begin.  The two
// have nothing to do with each other -- the push reads the edge of `inp`, the interior reads and writes elsewhere -- so here the
// first workgroups of the launch are the push (direct_block<U, true>, direct.hip.h), the rest the interior's strips
// (lap5_strip_tile, lap5.hip.h: the same tiles in the same XCD-aware order as lap5_strip_kernel, bit-identical results).
// The push workgroups are padded to a multiple of 8 so that the interior's workgroups keep their XCD (block % 8).
// Only for the 16-byte-lane paths of lap5_launch_variant (unit I stride, aligned rows); everything else keeps the two launches.
#pragma once

#include "direct.hip.h"
#include "lap5.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

// TPB tiles of 256 / TPB lanes per workgroup: rows of up to 64 / 128 lanes (local domains 128 / 256 columns wide for 8-byte
// items) fill the 256 threads the push role needs with 4 / 2 strips that follow each other along J.
template <typename T, typename W, int VARIANT, typename U, int TPB>
__global__ void __launch_bounds__(256)
lap5_interior_push_kernel(View<const T> in, View<T> out, int dI, int dJ, unsigned tiles_x, unsigned tiles_y, unsigned interior_tiles,
                          unsigned interior_blocks, unsigned push_pad, unsigned push_per_box, U* field, int64_t si, int64_t sj,
                          int64_t sk, BoxBatch b, DirectBatch d) {
    constexpr int VEC = 16 / (int)sizeof(T), LJ = Lap5Tuning::LJ, LANES = 256 / TPB;
    if (blockIdx.x < push_pad) {
        const unsigned m = blockIdx.x / push_per_box;
        if (m < (unsigned)b.n) direct_block<U, true>(field, si, sj, sk, b, d, (int)m, blockIdx.x % push_per_box);
        return;
    }
    const unsigned w = xcd_remap_grouped<(unsigned)Lap5Tuning::XCDG>(blockIdx.x - push_pad, interior_blocks);
    const unsigned t = w * TPB + threadIdx.x / LANES;  // (whole waves: LANES is a multiple of 64)
    if (t >= interior_tiles) return;
    const unsigned bx = t % tiles_x, by = (t / tiles_x) % tiles_y, k = t / (tiles_x * tiles_y);
    const unsigned lane = threadIdx.x & 63;
    int i0 = (int)(bx * LANES + threadIdx.x % LANES) * VEC;
    const bool active = i0 < dI;
    if (!active) i0 = dI - VEC;
    lap5_strip_lane<T, W, VARIANT, VEC, LJ>(in, out, dJ, i0, active, lane == 0, (lane == 63) || (i0 + VEC >= dI), (int)by * LJ, k, 0, dI);
}

// Interior [si, si + ei) x [sj, sj + ej) of `inp` -> `out` plus the push of `phase` of `plan`'s exchange of `inp`, one launch.
// *fused = false (and nothing launched) when the shapes do not qualify.
template <typename T, typename W>
inline int lap5_interior_with_push(gt4mi_halo_plan* plan, const int64_t sub[3], const gt4mi_field* a, const gt4mi_field* o, int variant,
                                   const gt4mi_field* exchanged, int phase, hipStream_t stream, bool* fused) {
    using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    *fused = false;
    if (sub[0] <= 0 || sub[1] <= 0 || sub[2] <= 0) return GT4MI_OK;
    constexpr int VMAX = 16 / (int)sizeof(T), LJ = Lap5Tuning::LJ;
    if (int rc = check_domain(sub)) return rc;
    const int h1[3] = {1, 1, 0}, h0[3] = {0, 0, 0};
    View<T> in_v, out_v;
    if (int rc = make_view<T>("inp", a, sub, h1, h1, &in_v)) return rc;
    if (int rc = make_view<T>("out", o, sub, h0, h0, &out_v)) return rc;
    if (views_overlap(in_v, h1, h1, out_v, h0, h0, sub))
        return fail(GT4MI_ERR_UNSUPPORTED, "lap5: 'inp' and 'out' overlap in memory (see gt4mi_lap5_*)");
    const View<const T> in_c{in_v.p, in_v.si, in_v.sj, in_v.sk};
    if (!(in_c.si == 1 && out_v.si == 1 && vec_ok(in_c, VMAX) && vec_ok(out_v, VMAX) && sub[0] % VMAX == 0))
        return GT4MI_OK;  // another kernel of lap5_launch_variant would run this interior: keep the two launches
    const int64_t lanes_per_row = sub[0] / VMAX;
    const int tpb = lanes_per_row <= 64 ? 4 : (lanes_per_row <= 128 ? 2 : 1);  // tiles of 64 / 128 / 256 lanes, as lap5_launch_variant
    const unsigned tx = (unsigned)cdiv(sub[0], (int64_t)(256 / tpb) * VMAX), ty = (unsigned)cdiv(sub[1], LJ);
    const int64_t tiles = (int64_t)tx * ty * sub[2], interior = cdiv(tiles, (int64_t)tpb);
    BoxBatch b;
    DirectBatch d;
    int64_t per_box = 0;
    if (int rc = direct_batches<U, true>(plan, exchanged, phase, b, d, per_box)) return rc;
    if (per_box == 0) return GT4MI_OK;  // nothing to push in this phase
    const int64_t pad = cdiv(per_box * b.n, (int64_t)8) * 8;
    if (tiles > INT32_MAX || interior + pad > INT32_MAX) return GT4MI_OK;
#define GT4MI_LAP5_PUSH_T(V, TPB)                                                                                                  \
    hipLaunchKernelGGL((lap5_interior_push_kernel<T, W, V, U, TPB>), dim3((unsigned)(interior + pad)), dim3(256),                   \
                       launch_dynamic_lds(), stream, in_c, out_v, (int)sub[0], (int)sub[1], tx, ty, (unsigned)tiles,                \
                       (unsigned)interior, (unsigned)pad, (unsigned)per_box, static_cast<U*>(exchanged->data),                      \
                       exchanged->stride[0] / (int64_t)sizeof(U), exchanged->stride[1] / (int64_t)sizeof(U),                        \
                       exchanged->stride[2] / (int64_t)sizeof(U), b, d)
#define GT4MI_LAP5_PUSH(V)                          \
    do {                                            \
        if (tpb == 1) GT4MI_LAP5_PUSH_T(V, 1);      \
        else if (tpb == 2) GT4MI_LAP5_PUSH_T(V, 2); \
        else GT4MI_LAP5_PUSH_T(V, 4);               \
    } while (0)
    switch (variant) {
        case GT4MI_LAP_NOTEBOOK: GT4MI_LAP5_PUSH(GT4MI_LAP_NOTEBOOK); break;
        case GT4MI_LAP_DOCS: GT4MI_LAP5_PUSH(GT4MI_LAP_DOCS); break;
        case GT4MI_LAP_SUITE: GT4MI_LAP5_PUSH(GT4MI_LAP_SUITE); break;
        case GT4MI_LAP_AVG: GT4MI_LAP5_PUSH(GT4MI_LAP_AVG); break;
        default: return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5: unknown variant %d", variant);
    }
#undef GT4MI_LAP5_PUSH
#undef GT4MI_LAP5_PUSH_T
    GT4MI_HIP_CHECK(hipGetLastError());
    *fused = true;
    return GT4MI_OK;
}

}  // namespace gt4mi
