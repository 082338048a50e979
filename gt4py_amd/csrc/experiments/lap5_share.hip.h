// (round 6 experiment, microbench section `lapshare`) The 5-point strip kernel with the J halo rows of neighbouring waves exchanged
// through LDS, as horizontal diffusion does since this round (../hdiff_share.hip.h): a workgroup is NWI waves side by side along I
// times NWJ blocks of LJ rows stacked along J; every wave loads its own LJ rows, publishes the first and the last one in LDS (2 KiB per
// wave), one barrier, and takes the row above / below its block from the wave above / below.  Row loads per workgroup column:
// NWJ * LJ + 2 instead of NWJ * (LJ + 2).
#pragma once

#include "common.hip.h"
#include "lane_shift.hip.h"
#include "lap5.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

template <typename T, typename W, int VARIANT, int VEC, int LJ, int NWI, int NWJ, int XCDG>
__global__ void __launch_bounds__(64 * NWI * NWJ)
lap5_share_kernel(View<const T> in, View<T> out, int dI, int dJ, unsigned tiles_x, unsigned tiles_y) {
    static_assert(VEC * sizeof(T) == 16, "16-byte lanes");
    using V = typename VecT<T, VEC>::type;
    __shared__ __attribute__((aligned(16))) char rows_lds[NWI * NWJ * 2 * 1024];
    unsigned b = blockIdx.x;
    if constexpr (XCDG > 0) b = xcd_remap_grouped<(unsigned)XCDG>(b, gridDim.x);
    const unsigned bx = b % tiles_x, by = (b / tiles_x) % tiles_y, k = b / (tiles_x * tiles_y);
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wi = wv % NWI, wj = wv / NWI;
    int i0 = (int)((bx * NWI + wi) * 64 + lane) * VEC;
    const bool active = i0 < dI;
    if (!active) i0 = dI - VEC;
    const bool edge_w = lane == 0, edge_e = (lane == 63) || (i0 + VEC >= dI);
    const int j0 = ((int)by * NWJ + wj) * LJ;

    const T* __restrict__ col = in.p + (int64_t)k * in.sk + i0;
    T* __restrict__ ocol = out.p + (int64_t)k * out.sk + i0;
    T r[LJ + 2][VEC];
    int64_t roff[LJ + 2];
#pragma unroll
    for (int t = 0; t < LJ + 2; ++t) {
        int jr = j0 - 1 + t;
        jr = jr > dJ ? dJ : jr;
        roff[t] = (int64_t)jr * in.sj;
    }
    // the rows the neighbours wait for first
    vload<T, VEC>(col + roff[1], r[1]);
    vload<T, VEC>(col + roff[LJ], r[LJ]);
#pragma unroll
    for (int t = 2; t < LJ; ++t) vload<T, VEC>(col + roff[t], r[t]);
    if (wj == 0) vload<T, VEC>(col + roff[0], r[0]);
    if (wj == NWJ - 1) vload<T, VEC>(col + roff[LJ + 1], r[LJ + 1]);
    auto slot = [&](int wave, int which) { return rows_lds + (wave * 2 + which) * 1024 + lane * 16; };
    {
        V a, z;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            a[q] = r[1][q];
            z[q] = r[LJ][q];
        }
        *reinterpret_cast<V*>(slot(wv, 0)) = a;
        *reinterpret_cast<V*>(slot(wv, 1)) = z;
    }
    __syncthreads();
    if (wj > 0) {
        const V v = *reinterpret_cast<const V*>(slot(wv - NWI, 1));
#pragma unroll
        for (int q = 0; q < VEC; ++q) r[0][q] = v[q];
    }
    if (wj < NWJ - 1) {
        const V v = *reinterpret_cast<const V*>(slot(wv + NWI, 0));
#pragma unroll
        for (int q = 0; q < VEC; ++q) r[LJ + 1][q] = v[q];
    }
    T w[LJ + 2], e[LJ + 2];
#pragma unroll
    for (int t = 1; t <= LJ; ++t) {
        T wl = lane_shift<T, true>(r[t][VEC - 1]);
        T el = lane_shift<T, false>(r[t][0]);
        if (edge_w) wl = col[roff[t] - 1];
        if (edge_e) el = col[roff[t] + VEC];
        w[t] = wl;
        e[t] = el;
    }
#pragma unroll
    for (int t = 1; t <= LJ; ++t) {
        T res[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const T wv_ = (v == 0) ? w[t] : r[t][v - 1];
            const T ev = (v == VEC - 1) ? e[t] : r[t][v + 1];
            res[v] = lap5_expr<T, W, VARIANT>(r[t][v], wv_, ev, r[t - 1][v], r[t + 1][v]);
        }
        if (active && (j0 + t - 1 < dJ)) vstore<T, VEC, true>(ocol + (int64_t)(j0 + t - 1) * out.sj, res);
    }
}

template <typename T, typename W, int VARIANT, int VEC, int LJ, int NWI, int NWJ, int XCDG>
inline void lap5_share_launch(const View<const T>& in, const View<T>& out, int dI, int dJ, int dK, hipStream_t stream) {
    const unsigned tx = (unsigned)cdiv(dI, NWI * 64 * VEC), ty = (unsigned)cdiv(dJ, LJ * NWJ);
    hipLaunchKernelGGL((lap5_share_kernel<T, W, VARIANT, VEC, LJ, NWI, NWJ, XCDG>), dim3(tx * ty * (unsigned)dK), dim3(64 * NWI * NWJ), 0, stream,
                       in, out, dI, dJ, tx, ty);
}

// (round 6 experiment, microbench section `lappersist`) the library's strip tiles, but a PERSISTENT grid: `grid` workgroups loop over the
// tiles (tile = blockIdx.x, += gridDim.x) instead of one workgroup per tile -- does the dispatch of 8 192 (512 x 512 x 128) or 32 768 (512^3)
// short-lived workgroups cost anything?  gridDim.x is a multiple of 8, so a workgroup's tiles stay on its XCD under the grouped remap.
template <typename T, typename W, int VARIANT, int VEC, int LJ, int BLOCK, int XCDG>
__global__ void __launch_bounds__(BLOCK)
lap5_persistent_kernel(View<const T> in, View<T> out, int dI, int dJ, unsigned tiles_x, unsigned tiles_y, unsigned n_tiles) {
    for (unsigned t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        unsigned b = t;
        if constexpr (XCDG > 0) b = xcd_remap_grouped<(unsigned)XCDG>(b, n_tiles);
        const unsigned bx = b % tiles_x, by = (b / tiles_x) % tiles_y, k = b / (tiles_x * tiles_y);
        lap5_strip_tile<T, W, VARIANT, VEC, LJ, BLOCK>(in, out, dI, dJ, bx, (int)by * LJ, k);
    }
}

}  // namespace gt4mi
