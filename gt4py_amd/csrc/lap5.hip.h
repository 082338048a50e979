// 5-point star stencils for gfx950: out[i,j,k] = f(in[i,j,k], in[i-1,j,k], in[i+1,j,k], in[i,j-1,k], in[i,j+1,k]).
//
// Reference semantics: the single-statement PARALLEL stencils listed in include/gt4py_amd.h
// (numpy backend code shape: /root/reference/src/gt4py/cartesian/gtc/numpy/npir_codegen.py:205-212).
//
// Roofline: HBM.  Algorithmic traffic 2*sizeof(T) bytes per lattice update (one read, one write).
//
// Fast path ("J-march"): a workgroup owns a strip BLOCK*VEC points wide in I (contiguous, one
// 16-byte vector per lane) and LJ rows tall in J at one K level.  Every lane walks down J keeping a
// three-row window (j-1, j, j+1) in registers, so each element of `in` is fetched from global
// memory once per strip (plus the two halo rows of the strip, 2/LJ extra).  The I-neighbours
// in[i-1], in[i+VEC] are one extra 4/8-byte load each that hits the line the neighbouring lane's
// vector load just brought into L1 -- no LDS, no barriers, waves run fully decoupled.
#pragma once

#include "common.hip.h"

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

template <typename T, typename W, int VARIANT, int VEC, int LJ, int BLOCK, bool NT, bool XCD>
__global__ void __launch_bounds__(BLOCK)
lap5_jmarch_kernel(View<const T> in, View<T> out, int dI, int dJ, unsigned tiles_x,
                   unsigned tiles_y, unsigned ntiles) {
    unsigned b = blockIdx.x;
    if constexpr (XCD) b = xcd_remap(b, ntiles);
    const unsigned bx = b % tiles_x;
    const unsigned by = (b / tiles_x) % tiles_y;
    const unsigned k = b / (tiles_x * tiles_y);

    const int i0 = (int)(bx * BLOCK + threadIdx.x) * VEC;
    if (i0 >= dI) return;
    const int j0 = (int)by * LJ;

    const T* __restrict__ row = in.p + (int64_t)k * in.sk + (int64_t)(j0 - 1) * in.sj + i0;
    T* __restrict__ orow = out.p + (int64_t)k * out.sk + (int64_t)j0 * out.sj + i0;

    T prev[VEC], cur[VEC], nxt[VEC];
    T cw, ce, nw = 0, ne = 0;
    vload<T, VEC>(row, prev);
    row += in.sj;
    vload<T, VEC>(row, cur);
    cw = row[-1];
    ce = row[VEC];

    auto step = [&](bool more) {
        row += in.sj;
        vload<T, VEC>(row, nxt);
        if (more) {
            nw = row[-1];
            ne = row[VEC];
        }
        T res[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const T w = (e == 0) ? cw : cur[e - 1];
            const T ee = (e == VEC - 1) ? ce : cur[e + 1];
            res[e] = lap5_expr<T, W, VARIANT>(cur[e], w, ee, prev[e], nxt[e]);
        }
        vstore<T, VEC, NT>(orow, res);
        orow += out.sj;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            prev[e] = cur[e];
            cur[e] = nxt[e];
        }
        cw = nw;
        ce = ne;
    };

    if (j0 + LJ <= dJ) {
#pragma unroll
        for (int jj = 0; jj < LJ; ++jj) step(jj + 1 < LJ);
    } else {
        const int nrows = dJ - j0;
        for (int jj = 0; jj < nrows; ++jj) step(jj + 1 < nrows);
    }
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
    static constexpr int LJ = 32;       // rows per strip (halo overhead 2/LJ)
    static constexpr bool NT = true;    // non-temporal stores: `out` is never re-read by this kernel
    static constexpr bool XCD = true;   // XCD-contiguous tile order
};

template <typename T, typename W, int VARIANT, int VEC, int BLOCK>
inline int lap5_launch_jmarch(const View<const T>& in, const View<T>& out, const int64_t d[3],
                              hipStream_t stream) {
    constexpr int LJ = Lap5Tuning::LJ;
    const unsigned tx = (unsigned)cdiv(d[0], (int64_t)BLOCK * VEC);
    const unsigned ty = (unsigned)cdiv(d[1], LJ);
    const int64_t n = (int64_t)tx * ty * d[2];
    if (n > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "lap5: domain too large for one launch");
    hipLaunchKernelGGL((lap5_jmarch_kernel<T, W, VARIANT, VEC, LJ, BLOCK, Lap5Tuning::NT, Lap5Tuning::XCD>),
                       dim3((unsigned)n), dim3(BLOCK), 0, stream, in, out, (int)d[0], (int)d[1], tx,
                       ty, (unsigned)n);
    return GT4MI_OK;
}

template <typename T, typename W, int VARIANT>
inline int lap5_launch_variant(const View<const T>& in, const View<T>& out, const int64_t d[3],
                               hipStream_t stream) {
    constexpr int VMAX = 16 / sizeof(T);
    if (in.si == 1 && out.si == 1) {
        const bool vec = vec_ok(in, VMAX) && vec_ok(out, VMAX) && (d[0] % VMAX == 0);
        if (vec) {
            const int64_t lanes = d[0] / VMAX;
            if (lanes <= 64) return lap5_launch_jmarch<T, W, VARIANT, VMAX, 64>(in, out, d, stream);
            if (lanes <= 128) return lap5_launch_jmarch<T, W, VARIANT, VMAX, 128>(in, out, d, stream);
            return lap5_launch_jmarch<T, W, VARIANT, VMAX, 256>(in, out, d, stream);
        }
        if (d[0] <= 64) return lap5_launch_jmarch<T, W, VARIANT, 1, 64>(in, out, d, stream);
        return lap5_launch_jmarch<T, W, VARIANT, 1, 256>(in, out, d, stream);
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

}  // namespace gt4mi
