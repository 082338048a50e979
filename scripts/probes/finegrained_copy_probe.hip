// How fast are 16-byte copies into / out of FINE-GRAINED device memory, with plain and with write-through / cache-bypassing
// (sc0 sc1) accesses?  0.5 MB and 2 MB by 128 workgroups, idle device.  build: hipcc --offload-arch=gfx950 -O2 <this> -o <exe>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int STORE, int LOAD>
__global__ void __launch_bounds__(256) copy(const u32x4* src, u32x4* dst, long n) {
    for (long t0 = (long)blockIdx.x * 1024 + threadIdx.x; t0 < n; t0 += (long)gridDim.x * 1024) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long t = t0 + u * 256 < n ? t0 + u * 256 : n - 1;
            if (LOAD == 0) v[u] = src[t];
            else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(src + t) : "memory");
        }
        if (LOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long t = t0 + u * 256;
            if (t < n) {
                if (STORE == 0) dst[t] = v[u];
                else if (STORE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + t), "v"(v[u]) : "memory");
                else if (STORE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + t), "v"(v[u]) : "memory");
                else __builtin_nontemporal_store(v[u], dst + t);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int STORE, int LOAD>
float run(const u32x4* src, u32x4* dst, long n, int blocks) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((copy<STORE, LOAD>), dim3(blocks), dim3(256), 0, 0, src, dst, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((copy<STORE, LOAD>), dim3(blocks), dim3(256), 0, 0, src, dst, n);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 50 * 1000;
}

int main() {
    for (long bytes : {524288L, 2097152L}) {
        const long n = bytes / 16;
        const int blocks = (int)((n + 1023) / 1024);
        u32x4 *coarse_a, *coarse_b, *fine;
        CK(hipMalloc((void**)&coarse_a, bytes)); CK(hipMalloc((void**)&coarse_b, bytes));
        CK(hipExtMallocWithFlags((void**)&fine, bytes, hipDeviceMallocFinegrained));
        CK(hipMemset(coarse_a, 1, bytes)); CK(hipMemset(fine, 2, bytes));
        printf("%ld bytes, %d workgroups, us per copy:\n", bytes, blocks);
        printf("  coarse -> coarse, plain loads, plain stores        %7.1f\n", run<0, 0>(coarse_a, coarse_b, n, blocks));
        printf("  coarse -> fine,   plain loads, plain stores        %7.1f\n", run<0, 0>(coarse_a, fine, n, blocks));
        printf("  coarse -> fine,   plain loads, sc0 sc1 stores      %7.1f\n", run<1, 0>(coarse_a, fine, n, blocks));
        printf("  coarse -> fine,   plain loads, sc1 stores          %7.1f\n", run<2, 0>(coarse_a, fine, n, blocks));
        printf("  coarse -> fine,   plain loads, nontemporal stores  %7.1f\n", run<3, 0>(coarse_a, fine, n, blocks));
        printf("  coarse -> coarse, plain loads, sc0 sc1 stores      %7.1f\n", run<1, 0>(coarse_a, coarse_b, n, blocks));
        printf("  fine -> coarse,   plain loads, plain stores        %7.1f\n", run<0, 0>(fine, coarse_b, n, blocks));
        printf("  fine -> coarse,   sc0 sc1 loads, plain stores      %7.1f\n", run<0, 1>(fine, coarse_b, n, blocks));
        printf("  coarse -> coarse, sc0 sc1 loads, plain stores      %7.1f\n", run<0, 1>(coarse_a, coarse_b, n, blocks));
        CK(hipFree(coarse_a)); CK(hipFree(coarse_b)); CK(hipFree(fine));
    }
    return 0;
}
