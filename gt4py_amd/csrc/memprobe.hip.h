// Which MEMORY GROUP does an allocation live in?  (round 5; profiles/r5_memory_groups.txt)
//
// Measured on MI355X: the device's memory is not one uniformly interleaved pool.  Two big allocations either share a group of memory
// channels or they do not, and kernels feel it: writing two 1.3 GB fields side by side runs at 5.0-6.4 TB/s when both live in the
// same group and at 6.8-7.0 TB/s when they live in different ones; the fp64 Laplacian on 512^3 gains 2.3 % with `in` and `out` in
// different groups, the tridiagonal solve runs at 0.70 of the HBM peak with its five fields dealt over two groups and at 0.61 with
// all five in one -- the "two speed modes by allocation set" of rounds 2-4.  Nothing in the API says which group an allocation got;
// this probe measures it: ONE kernel writes both buffers the way a column kernel does (a wave per 64 columns of a row, level after
// level, a plane of 8 MiB apart, 512 bytes per level and buffer) and the bandwidth of the pair tells same group from different groups.
//
// The probe OVERWRITES the first `bytes` of both buffers: it is for fresh allocations and for buffers the allocator owns
// (gt4py_amd/storage/placement.py), never for user data.
#pragma once

#include "common.hip.h"

namespace gt4mi {

constexpr int64_t PROBE_ROW = 1024, PROBE_PLANE = PROBE_ROW * 1024;  // items of 8 bytes: rows of 8 KiB, planes of 8 MiB

// One wave: 64 columns of one row of every plane; `b` may be null (one buffer alone).
__global__ void __launch_bounds__(64) memory_write_probe_kernel(double* __restrict__ a, double* __restrict__ b, int levels) {
    const unsigned tile = blockIdx.x % 16u, row = blockIdx.x / 16u;
    const int64_t off = (int64_t)row * PROBE_ROW + (int64_t)tile * 64 + threadIdx.x;
    if (b != nullptr) {
#pragma unroll 8
        for (int k = 0; k < levels; ++k) {
            a[off + (int64_t)k * PROBE_PLANE] = 4.5;
            b[off + (int64_t)k * PROBE_PLANE] = 5.5;
        }
    } else {
#pragma unroll 8
        for (int k = 0; k < levels; ++k) a[off + (int64_t)k * PROBE_PLANE] = 4.5;
    }
}

// GB/s (bytes written to BOTH buffers per second) of `iterations` launches after 2 warm-up launches; synchronous (HIP events on
// `stream`, waited for).  bytes is rounded down to whole planes; at least 24 planes (192 MiB).
inline int memory_write_probe(void* a, void* b, size_t bytes, int iterations, hipStream_t stream, double* gbs) {
    if (a == nullptr || gbs == nullptr) return fail(GT4MI_ERR_INVALID_ARGUMENT, "memory_write_probe: null pointer");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 8 != 0)
        return fail(GT4MI_ERR_INVALID_ARGUMENT, "memory_write_probe: buffers must be 8-byte aligned");
    // both buffers must live on the device the stream (= the current device: one process per GPU, one placer per device) belongs
    // to: a kernel launched on device A that writes memory of device B faults unless peer access happens to be enabled
    {
        int current = -1;
        GT4MI_HIP_CHECK(hipGetDevice(&current));
        for (void* p : {a, b}) {
            if (p == nullptr) continue;
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
                (void)hipGetLastError();
                return fail(GT4MI_ERR_INVALID_ARGUMENT, "memory_write_probe: %p is not a device allocation", p);
            }
            if (attr.device != current)
                return fail(GT4MI_ERR_INVALID_ARGUMENT, "memory_write_probe: buffer %p lives on device %d, the current device is %d", p, attr.device, current);
        }
    }
    const int64_t levels = (int64_t)(bytes / (PROBE_PLANE * sizeof(double)));
    // (a launch writes the span of one or two buffers front to back, launch after launch: once the span exceeds the 256 MB Infinity
    // Cache every write evicts a dirty line and the memory sees all of it; 24 planes = 192 MiB per buffer is the floor for a PAIR)
    if (levels < 24) return fail(GT4MI_ERR_INVALID_ARGUMENT, "memory_write_probe: %zu bytes are fewer than 24 planes of 8 MiB (the Infinity Cache would absorb the writes)", bytes);
    if (levels > INT32_MAX) return fail(GT4MI_ERR_INVALID_ARGUMENT, "memory_write_probe: too many planes");
    if (iterations < 1) iterations = 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    GT4MI_HIP_CHECK(hipEventCreate(&e0));
    GT4MI_HIP_CHECK(hipEventCreate(&e1));
    auto launch = [&]() {
        hipLaunchKernelGGL(memory_write_probe_kernel, dim3(16u * 1024u), dim3(64), 0, stream, static_cast<double*>(a), static_cast<double*>(b), (int)levels);
    };
    for (int i = 0; i < 2; ++i) launch();
    hipError_t rc = hipEventRecord(e0, stream);
    for (int i = 0; i < iterations && rc == hipSuccess; ++i) launch();
    if (rc == hipSuccess) rc = hipEventRecord(e1, stream);
    if (rc == hipSuccess) rc = hipEventSynchronize(e1);
    float ms = 0.f;
    if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != hipSuccess) return fail(GT4MI_ERR_HIP, "memory_write_probe: %s", hipGetErrorString(rc));
    GT4MI_HIP_CHECK(hipGetLastError());
    const double written = (double)levels * PROBE_PLANE * sizeof(double) * (b != nullptr ? 2.0 : 1.0) * iterations;
    *gbs = ms > 0.f ? written / (ms * 1e-3) / 1e9 : 0.0;
    return GT4MI_OK;
}

}  // namespace gt4mi
