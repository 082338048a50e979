// Horizontal diffusion (lap-of-lap with optional flux limiter) for gfx950.
//
// Reference semantics: `horizontal_diffusion` / `simple_horizontal_diffusion`
//   /root/reference/tests/cartesian_tests/integration_tests/multi_feature_tests/stencil_definitions.py:316-328, :206-216
// evaluated statement by statement as the numpy backend does:
//   lap = 4.0*in - (((in[1,0] + in[-1,0]) + in[0,1]) + in[0,-1])
//   res = lap[1,0] - lap ; flx = (res*(in[1,0]-in) > 0) ? 0 : res
//   res = lap[0,1] - lap ; fly = (res*(in[0,1]-in) > 0) ? 0 : res
//   out = in - coeff*(((flx - flx[-1,0]) + fly) - fly[0,-1])
// Every intermediate is a pure function of `in`, so recomputing it per consumer is value-identical
// to storing it in a temporary (no FMA, no reassociation).
//
// dtype rules (gtir_upcaster.py:43-143): T = field dtype, W = dtype of lap/flx/fly (double for
// float fields under the default float64 literals), PW = dtype of coeff*(...) and of the final
// subtraction (double whenever W or a scalar coeff is double).
//
// Roofline: HBM. Algorithmic traffic 3*sizeof(T) bytes per lattice update with a coefficient field
// (in, coeff read; out written), 2*sizeof(T) with a scalar coefficient.
#pragma once

#include "common.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

// lap at a point given its five `in` values (c, +i, -i, +j, -j).
template <typename T, typename W>
__device__ __forceinline__ W hd_lap(T c, T ip, T im, T jp, T jm) {
    const T sum = ((ip + im) + jp) + jm;
    return ((W)4.0 * (W)c) - (W)sum;
}

// flux between a point (lap0, in0) and its +1 neighbour (lap1, in1).
template <typename T, typename W, bool LIMITER>
__device__ __forceinline__ W hd_flux(W lap1, W lap0, T in1, T in0) {
    const W res = lap1 - lap0;
    if constexpr (LIMITER) {
        const W d = (W)(in1 - in0);
        return ((res * d) > (W)0) ? (W)0 : res;
    } else {
        return res;
    }
}

template <typename T, typename W, typename PW>
__device__ __forceinline__ T hd_out(T in0, PW coeff, W flx, W flxm, W fly, W flym) {
    const W s = ((flx - flxm) + fly) - flym;
    return (T)((PW)in0 - (coeff * (PW)s));
}

// ---------------------------------------------------------------------------------------------
// Any-stride kernel: one thread per (i,j) point, K across gridDim.z.  13 loads of `in` per point,
// served by L1/L2 for the overlapping neighbourhoods.  Used for non-I-contiguous layouts and as
// the in-library cross-check of the J-march kernel.
// ---------------------------------------------------------------------------------------------
// TRANSPOSED = true: lanes run along J and the 4 sub-rows of a workgroup along I -- for domains only a few
// columns wide (the west / east boundary strips of an IJ-decomposed apply), where lanes along I would
// leave 62 of 64 lanes idle.  Same arithmetic per point.
template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD, bool TRANSPOSED = false>
__global__ void __launch_bounds__(256)
hdiff_generic_kernel(View<const T> in, View<T> out, View<const T> cf, PW coeff_scalar, int dI,
                     int dJ, int dK) {
    const int a = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int i = TRANSPOSED ? b : a;
    const int j = TRANSPOSED ? a : b;
    if (i >= dI || j >= dJ) return;
    const int64_t si = in.si, sj = in.sj;
    for (int k = blockIdx.z; k < dK; k += gridDim.z) {
        const T* p = in.p + (int64_t)i * si + (int64_t)j * sj + (int64_t)k * in.sk;
        const T c = p[0];
        const T e1 = p[si], e2 = p[2 * si], w1 = p[-si], w2 = p[-2 * si];
        const T n1 = p[sj], n2 = p[2 * sj], s1 = p[-sj], s2 = p[-2 * sj];
        const T ne = p[si + sj], nw = p[-si + sj], se = p[si - sj], sw = p[-si - sj];
        const W lap_c = hd_lap<T, W>(c, e1, w1, n1, s1);
        const W lap_e = hd_lap<T, W>(e1, e2, c, ne, se);
        const W lap_w = hd_lap<T, W>(w1, c, w2, nw, sw);
        const W lap_n = hd_lap<T, W>(n1, ne, nw, n2, c);
        const W lap_s = hd_lap<T, W>(s1, se, sw, c, s2);
        const W flx = hd_flux<T, W, LIMITER>(lap_e, lap_c, e1, c);
        const W flxm = hd_flux<T, W, LIMITER>(lap_c, lap_w, c, w1);
        const W fly = hd_flux<T, W, LIMITER>(lap_n, lap_c, n1, c);
        const W flym = hd_flux<T, W, LIMITER>(lap_c, lap_s, c, s1);
        PW coeff;
        if constexpr (COEFF_FIELD)
            coeff = (PW)cf.p[(int64_t)i * cf.si + (int64_t)j * cf.sj + (int64_t)k * cf.sk];
        else
            coeff = coeff_scalar;
        out.p[(int64_t)i * out.si + (int64_t)j * out.sj + (int64_t)k * out.sk] =
            hd_out<T, W, PW>(c, coeff, flx, flxm, fly, flym);
    }
}

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD>
inline int hdiff_launch(const View<const T>& in, const View<T>& out, const View<const T>& cf,
                        PW coeff_scalar, const int64_t d[3], hipStream_t stream, bool point_per_thread = false);

}  // namespace gt4mi

#include "hdiff_jmarch.hip.h"
#include "hdiff_share.hip.h"

namespace gt4mi {

template <typename T, typename W, typename PW, bool LIMITER, bool COEFF_FIELD>
inline int hdiff_launch(const View<const T>& in, const View<T>& out, const View<const T>& cf,
                        PW coeff_scalar, const int64_t d[3], hipStream_t stream, bool point_per_thread) {
    // point_per_thread: `coeff` IS `out` (same elements): hdiff_generic_kernel reads a point's coefficient and
    // then writes that point, nothing else touches the element and no pointer is declared __restrict__
    const bool contiguous = !point_per_thread && in.si == 1 && out.si == 1 && (!COEFF_FIELD || cf.si == 1);
    // Domains only a few columns wide (the west / east boundary strips of an IJ-decomposed apply) would use
    // 2 of the 128-256 columns of a wave-wide tile: one thread per point is 5-10x faster there.
    const bool skinny = d[0] < 32;
    if (contiguous && !skinny && hdiff_jmarch_enabled()) {
        // 16-byte lanes: the J-march whose waves exchange their halo rows through LDS (hdiff_share.hip.h, round 6);
        // GT4MI_HDIFF_SHARE=0 keeps the register-only J-march of rounds 1-5 for A/B runs on ONE box
        constexpr int VMAX = 16 / sizeof(T);
        static const bool share = env_int("GT4MI_HDIFF_SHARE", 1) != 0;
        int lead = 0;
        if (share && hdiff_common_lead<T, COEFF_FIELD>(in, out, cf, VMAX, &lead))
            return hdiff_launch_share<T, W, PW, LIMITER, COEFF_FIELD, VMAX>(in, out, cf, coeff_scalar, d, stream, lead);
        return hdiff_launch_jmarch<T, W, PW, LIMITER, COEFF_FIELD>(in, out, cf, coeff_scalar, d, stream);
    }
    if (skinny) {
        dim3 grid((unsigned)cdiv(d[1], 64), (unsigned)cdiv(d[0], 4), (unsigned)(d[2] < 65535 ? d[2] : 65535));
        hipLaunchKernelGGL((hdiff_generic_kernel<T, W, PW, LIMITER, COEFF_FIELD, true>), grid, dim3(256), 0, stream, in,
                           out, cf, coeff_scalar, (int)d[0], (int)d[1], (int)d[2]);
        return GT4MI_OK;
    }
    dim3 grid((unsigned)cdiv(d[0], 64), (unsigned)cdiv(d[1], 4),
              (unsigned)(d[2] < 65535 ? d[2] : 65535));
    hipLaunchKernelGGL((hdiff_generic_kernel<T, W, PW, LIMITER, COEFF_FIELD>), grid, dim3(256), 0,
                       stream, in, out, cf, coeff_scalar, (int)d[0], (int)d[1], (int)d[2]);
    return GT4MI_OK;
}

template <typename T, typename W, typename PW>
inline int hdiff_dispatch(const View<const T>& in, const View<T>& out, const View<const T>& cf,
                          bool coeff_field, PW coeff_scalar, bool limiter, const int64_t d[3],
                          hipStream_t stream, bool point_per_thread = false) {
    if (limiter) {
        if (coeff_field) return hdiff_launch<T, W, PW, true, true>(in, out, cf, coeff_scalar, d, stream, point_per_thread);
        return hdiff_launch<T, W, PW, true, false>(in, out, cf, coeff_scalar, d, stream, point_per_thread);
    }
    if (coeff_field) return hdiff_launch<T, W, PW, false, true>(in, out, cf, coeff_scalar, d, stream, point_per_thread);
    return hdiff_launch<T, W, PW, false, false>(in, out, cf, coeff_scalar, d, stream, point_per_thread);
}

template <typename T>
inline int hdiff_run(const int64_t domain[3], const gt4mi_field* in_field,
                     const gt4mi_field* out_field, const gt4mi_field* coeff, double coeff_scalar,
                     int flags, hipStream_t stream) {
    if (int rc = check_domain(domain)) return rc;
    const int h2[3] = {2, 2, 0}, h0[3] = {0, 0, 0};
    View<T> in_v, out_v, cf_v{nullptr, 0, 0, 0};
    if (int rc = make_view<T>("in_field", in_field, domain, h2, h2, &in_v)) return rc;
    if (int rc = make_view<T>("out_field", out_field, domain, h0, h0, &out_v)) return rc;
    if (coeff != nullptr)
        if (int rc = make_view<T>("coeff", coeff, domain, h0, h0, &cf_v)) return rc;
    if (domain[0] == 0 || domain[1] == 0 || domain[2] == 0) return GT4MI_OK;
    if (views_overlap(in_v, h2, h2, out_v, h0, h0, domain))
        return fail(GT4MI_ERR_UNSUPPORTED,
                    "hdiff: 'in_field' and 'out_field' overlap in memory; every point reads its neighbours' OLD values "
                    "(the reference evaluates the right-hand side before it assigns), which an in-place kernel cannot "
                    "provide -- pass a separate output array");
    bool alias = false;  // out_field IS coeff: a point's coefficient is read before that point is written
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
        double cs = (flags & GT4MI_HDIFF_COEFF_F32) ? (double)(float)coeff_scalar : coeff_scalar;
        rc = hdiff_dispatch<T, double, double>(in_c, out_v, cf_c, has_field, cs, limiter, domain, stream, alias);
    } else {
        const bool w32 = (flags & GT4MI_HDIFF_INTERNAL_F32) != 0;
        const bool c32 = (flags & GT4MI_HDIFF_COEFF_F32) != 0;
        if (!w32) {
            double cs = c32 ? (double)(float)coeff_scalar : coeff_scalar;
            rc = hdiff_dispatch<T, double, double>(in_c, out_v, cf_c, has_field, cs, limiter, domain, stream, alias);
        } else if (has_field || c32) {
            rc = hdiff_dispatch<T, float, float>(in_c, out_v, cf_c, has_field, (float)coeff_scalar, limiter, domain, stream, alias);
        } else {
            rc = hdiff_dispatch<T, float, double>(in_c, out_v, cf_c, has_field, coeff_scalar, limiter, domain, stream, alias);
        }
    }
    if (rc) return rc;
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

}  // namespace gt4mi
