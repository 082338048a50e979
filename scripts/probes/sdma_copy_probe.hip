// VERDICT round 4, item 5 -- ONE bounded experiment before the RCCL path is frozen: could the packed faces travel by a copy ENGINE
// (SDMA) instead of a CU kernel?  RCCL's send/recv kernel takes 59-63 us next to an HBM-saturating interior kernel (13 us alone,
// profiles/r4_dist_lap5_apply_timeline_share_4x2_swap-packed_rccl.txt): it starves for memory bandwidth and CU slots.  A copy
// engine needs neither a CU nor a wave slot.  This probe measures exactly the step a "copy" transport would put in RCCL's place:
//
//   side stream:  N asynchronous device-to-device copies of the faces of a 128 x 256 x 512 share (2 x 1.05 MB + 2 x 0.53 MB) into a
//                 fine-grained pool (what the direct transport exports over hipIpc), then a ONE-WAVE kernel that raises the flags
//   main stream:  an HBM-saturating streaming kernel (the stand-in for the interior kernel), or nothing
//
// for three kinds of copy: hipMemcpyDeviceToDevice (the runtime picks: a blit KERNEL on one device), hipMemcpyDeviceToDeviceNoCU
// (the runtime must not use compute units: SDMA) and hipMemcpyPeerAsync(dev 0 -> dev 0).  Reported: microseconds from the first
// copy's issue to the flag kernel's completion (HIP events on the side stream), idle and under load, median of 50.
// Under rocprofv3 --kernel-trace --memory-copy-trace the copies show up as kernels (__amd_rocclr_copyBuffer) or as memory copies.
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/sdma_copy_probe.hip -o /tmp/sdma_copy_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) stream_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) __builtin_nontemporal_store(src[t], dst + t);
}

__global__ void raise_flags(unsigned* flags, int n) {
    if (threadIdx.x < (unsigned)n) __hip_atomic_fetch_add(flags + threadIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    const size_t faces[4] = {256 * 512 * 8, 256 * 512 * 8, 130 * 512 * 8, 130 * 512 * 8};
    char *send = nullptr, *pool = nullptr;
    size_t total = 4096;
    for (size_t f : faces) total += (f + 255) / 256 * 256;
    CK(hipMalloc(&send, total));
    CK(hipExtMallocWithFlags((void**)&pool, total, hipDeviceMallocFinegrained));
    CK(hipMemset(pool, 0, total));
    const long big = 1L << 30;  // 1 GiB each way: ~0.34 ms of saturating traffic per launch
    u32x4 *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, big));
    CK(hipMalloc(&b, big));
    hipStream_t main_s, side;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    struct Kind { const char* name; int how; };
    const Kind kinds[] = {{"hipMemcpyAsync DeviceToDevice       ", 0}, {"hipMemcpyAsync DeviceToDeviceNoCU   ", 1}, {"hipMemcpyPeerAsync (device 0 -> 0)  ", 2}};
    for (int loaded = 0; loaded < 2; ++loaded)
        for (const Kind& k : kinds) {
            std::vector<float> us;
            bool failed = false;
            for (int it = 0; it < 55 && !failed; ++it) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                if (loaded)
                    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(stream_copy, dim3(8192), dim3(256), 0, main_s, a, b, big / 16);
                CK(hipEventRecord(e0, side));
                size_t off = 4096;
                for (size_t f : faces) {
                    hipError_t rc = k.how == 0   ? hipMemcpyAsync(pool + off, send + off, f, hipMemcpyDeviceToDevice, side)
                                    : k.how == 1 ? hipMemcpyAsync(pool + off, send + off, f, hipMemcpyDeviceToDeviceNoCU, side)
                                                 : hipMemcpyPeerAsync(pool + off, 0, send + off, 0, f, side);
                    if (rc != hipSuccess) {
                        printf("%s %s: %s\n", k.name, loaded ? "under load" : "idle      ", hipGetErrorString(rc));
                        (void)hipGetLastError();
                        failed = true;
                        break;
                    }
                    off += (f + 255) / 256 * 256;
                }
                if (failed) break;
                hipLaunchKernelGGL(raise_flags, dim3(1), dim3(64), 0, side, (unsigned*)pool, 4);
                CK(hipEventRecord(e1, side));
                CK(hipEventSynchronize(e1));
                CK(hipStreamSynchronize(main_s));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 5) us.push_back(ms * 1000.f);
                CK(hipEventDestroy(e0));
                CK(hipEventDestroy(e1));
            }
            if (failed || us.empty()) continue;
            std::sort(us.begin(), us.end());
            printf("%s %s: 4 faces (3.2 MB) + flag kernel  median %7.1f us  min %7.1f  max %7.1f\n", k.name, loaded ? "under load" : "idle      ",
                   us[us.size() / 2], us.front(), us.back());
        }
    // the yardstick: the direct transport's own pack kernel moves the same bytes in 5-9 us idle; RCCL's send/recv kernel 13 us idle,
    // 59-63 us under load (profiles/r4_dist_lap5_apply_timeline_*)
    return 0;
}
