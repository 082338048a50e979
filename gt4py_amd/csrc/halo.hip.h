// Halo pack/unpack and the streaming-copy yardstick.
//
// NEW component: gt4py.cartesian has no multi-device path.  The ghost cells of a
// read field with a non-zero horizontal extent are exchanged between IJ-neighbour ranks as dense
// buffers (I fastest, then J, then K); these kernels gather/scatter one box of a strided field.
#pragma once

#include "common.hip.h"

namespace gt4mi {

template <typename U, bool PACK>
__global__ void __launch_bounds__(256)
halo_copy_kernel(U* field, int64_t si, int64_t sj, int64_t sk, U* buffer, int ei, int ej, int ek) {
    const int64_t n = (int64_t)ei * ej * ek;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int i = (int)(t % ei);
        const int64_t r = t / ei;
        const int j = (int)(r % ej);
        const int k = (int)(r / ej);
        U* f = field + i * si + j * sj + k * sk;
        if constexpr (PACK) buffer[t] = *f;
        else *f = buffer[t];
    }
}

template <typename U, bool PACK>
inline int halo_copy(const gt4mi_field* field, const int64_t lo[3], const int64_t ext[3], void* buffer,
                     hipStream_t stream) {
    if (field == nullptr || field->data == nullptr || buffer == nullptr || lo == nullptr || ext == nullptr)
        return fail(GT4MI_ERR_INVALID_ARGUMENT, "halo: null argument");
    int64_t off = 0;
    for (int a = 0; a < 3; ++a) {
        if (ext[a] < 0 || lo[a] < 0 || lo[a] + ext[a] > field->shape[a])
            return fail(GT4MI_ERR_OUT_OF_BOUNDS, "halo: box [%lld, %lld) outside of axis %d (size %lld)",
                        (long long)lo[a], (long long)(lo[a] + ext[a]), a, (long long)field->shape[a]);
        if (field->stride[a] % (int64_t)sizeof(U) != 0)
            return fail(GT4MI_ERR_UNSUPPORTED, "halo: stride not a multiple of the item size");
        if (ext[a] > INT32_MAX) return fail(GT4MI_ERR_UNSUPPORTED, "halo: box too large");
        off += lo[a] * field->stride[a];
    }
    const int64_t n = ext[0] * ext[1] * ext[2];
    if (n == 0) return GT4MI_OK;
    U* base = reinterpret_cast<U*>(static_cast<char*>(field->data) + off);
    int64_t blocks = cdiv(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((halo_copy_kernel<U, PACK>), dim3((unsigned)blocks), dim3(256), 0, stream, base,
                       field->stride[0] / (int64_t)sizeof(U), field->stride[1] / (int64_t)sizeof(U),
                       field->stride[2] / (int64_t)sizeof(U), static_cast<U*>(buffer), (int)ext[0],
                       (int)ext[1], (int)ext[2]);
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

// 16 bytes per lane, UNROLL independent vectors per thread, grid-stride.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256)
stream_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base < nvec; base += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (base + (size_t)u * 256 < nvec) {
                // (round 5) a copy reads every byte once: nontemporal loads too -- 6.29 -> 6.51 TB/s on one box, 6.67 with source and
                // destination in different memory groups (profiles/r5_nt_loads_column_kernels.txt, microbench `copynt`)
                if constexpr (NT) v[u] = __builtin_nontemporal_load(&src[base + (size_t)u * 256]);
                else v[u] = src[base + (size_t)u * 256];
            }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (base + (size_t)u * 256 < nvec) {
                if constexpr (NT) __builtin_nontemporal_store(v[u], &dst[base + (size_t)u * 256]);
                else dst[base + (size_t)u * 256] = v[u];
            }
    }
}

}  // namespace gt4mi
