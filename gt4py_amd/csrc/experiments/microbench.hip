// Stand-alone tuning harness for the gfx950 stencil kernels (not part of the product path).
// Prints one line per (kernel, configuration): ms per apply, GLUPS, algorithmic GB/s, % of 8 TB/s.
// Also self-checks: DPP wave shifts, J-march kernels against the any-stride kernels (bitwise).
//
//   microbench [section ...]     sections: dpp copy lap hdiff tridiag (default: all)
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hip.h"
#include "halo.hip.h"
#include "hdiff.hip.h"
#include "lap5.hip.h"
#include "tridiag.hip.h"
#include "vadv_stack.hip.h"
#include "hdiff_ldstile.hip.h"
#include "lap5_share.hip.h"

using namespace gt4mi;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

static constexpr double PEAK_GBS = 8000.0;

template <typename F>
static double time_ms(F&& launch, int iters, int warmup = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < warmup; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / iters;
}

static void report(const char* name, const char* cfg, double ms, double lups, double bytes_per_lup) {
    const double glups = lups / (ms * 1e-3) / 1e9;
    const double gbs = glups * bytes_per_lup;
    printf("%-10s %-44s %9.4f ms %9.1f GLUPS %9.1f GB/s %6.1f%% of 8TB/s\n", name, cfg, ms, glups, gbs,
           100.0 * gbs / PEAK_GBS);
    fflush(stdout);
}

// ---- a gt4py-storage-like padded field: I contiguous, rows padded to `align` items, element
// [halo] of every row aligned to align*sizeof(T) bytes -------------------------------------------
template <typename T>
struct DevField {
    char* raw = nullptr;
    T* data = nullptr;  // element [0,0,0]
    int64_t shape[3], sj, sk;
    int halo[3];
    size_t bytes;
    DevField(int64_t ni, int64_t nj, int64_t nk, int hi, int hj, int align_items = 32, int64_t extra_pitch = 0) {
        shape[0] = ni + 2 * hi;
        shape[1] = nj + 2 * hj;
        shape[2] = nk;
        halo[0] = hi;
        halo[1] = hj;
        halo[2] = 0;
        sj = cdiv(shape[0], align_items) * align_items + extra_pitch;
        sk = sj * shape[1];
        const size_t align_b = align_items * sizeof(T);
        bytes = (size_t)sk * nk * sizeof(T) + 2 * align_b;
        // MB_CONTIG=1: physically contiguous device memory (hipDeviceMallocContiguous) -- the placement experiment of round 4
        static const bool contiguous = getenv("MB_CONTIG") && atoi(getenv("MB_CONTIG")) != 0;
        // MB_VMM=<log2 of the VA alignment, e.g. 30>: the field is built with the virtual-memory API from physical chunks of
        // power-of-two sizes (largest first), each mapped at a VA offset that is a multiple of its size inside a reservation aligned
        // to 2^MB_VMM bytes -- if the physical allocator hands out naturally aligned chunks, VA and PA of every chunk are congruent
        // modulo the chunk size and the page tables can use the largest fragments (round 5: the allocation-class hypothesis)
        static const int vmm = getenv("MB_VMM") ? atoi(getenv("MB_VMM")) : 0;
        if (vmm > 0) {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            size_t gran = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            const size_t total = (bytes + gran - 1) / gran * gran;
            static const size_t max_chunk = getenv("MB_VMM_MAX_CHUNK_LOG2") ? (size_t)1 << atoi(getenv("MB_VMM_MAX_CHUNK_LOG2")) : (size_t)1 << 30;
            // (hipMemAddressReserve ignores its alignment argument on this runtime: reserve 2^MB_VMM more and start at the next multiple)
            void* reserved = nullptr;
            const size_t align = (size_t)1 << vmm;
            CK(hipMemAddressReserve(&reserved, total + align, align, nullptr, 0));
            vmm_reserved = (char*)reserved;
            vmm_reserved_bytes = total + align;
            void* va = (void*)(((uintptr_t)reserved + align - 1) / align * align);
            size_t off = 0;
            while (off < total) {
                size_t chunk = max_chunk;
                while (chunk > total - off || (off % chunk) != 0) chunk >>= 1;  // largest power of two that fits and keeps the offset aligned
                if (chunk < gran) chunk = gran;
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, chunk, &prop, 0));
                CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
                vmm_handles.push_back(h);
                off += chunk;
            }
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, total, &acc, 1));
            raw = (char*)va;
            vmm_total = total;
        } else if (contiguous) CK(hipExtMallocWithFlags((void**)&raw, bytes, hipDeviceMallocContiguous));
        else CK(hipMalloc(&raw, bytes));
        // offset so that element index `hi` is aligned
        const size_t off = (align_b - ((size_t)hi * sizeof(T)) % align_b) % align_b;
        data = reinterpret_cast<T*>(raw + off);
    }
    std::vector<hipMemGenericAllocationHandle_t> vmm_handles;
    size_t vmm_total = 0, vmm_reserved_bytes = 0;
    char* vmm_reserved = nullptr;
    ~DevField() {
        if (vmm_total) {
            (void)hipMemUnmap(raw, vmm_total);
            for (auto h : vmm_handles) (void)hipMemRelease(h);
            (void)hipMemAddressFree(vmm_reserved, vmm_reserved_bytes);
        } else {
            (void)hipFree(raw);
        }
    }
    View<T> view() const {  // origin-shifted
        return View<T>{data + halo[0] + halo[1] * sj, 1, sj, sk};
    }
    View<const T> cview() const {
        return View<const T>{data + halo[0] + halo[1] * sj, 1, sj, sk};
    }
    size_t elems() const { return (size_t)sk * shape[2]; }
};

template <typename T>
__global__ void fill_kernel(T* p, size_t n, unsigned seed, T lo, T hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ULL + seed * 0xD1B54A32D192ED03ULL;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ULL;
        x ^= x >> 32;
        const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
        p[i] = (T)(lo + (hi - lo) * u);
    }
}

template <typename T>
static void fill(DevField<T>& f, unsigned seed, double lo, double hi) {
    hipLaunchKernelGGL(fill_kernel<T>, dim3(4096), dim3(256), 0, 0, f.data, f.elems() - 64, seed, (T)lo, (T)hi);
    CK(hipDeviceSynchronize());
}

template <typename T>
__global__ void diff_kernel(View<const T> a, View<const T> b, int dI, int dJ, int dK, unsigned long long* nbad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y, k = blockIdx.z;
    if (i >= dI) return;
    const T x = a.p[i + j * a.sj + k * a.sk], y = b.p[i + j * b.sj + k * b.sk];
    const bool same = (x == y) || (x != x && y != y);
    if (!same) atomicAdd(nbad, 1ULL);
}

template <typename T>
static unsigned long long count_diff(const DevField<T>& a, const DevField<T>& b, int dI, int dJ, int dK) {
    unsigned long long* d;
    CK(hipMalloc(&d, 8));
    CK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(diff_kernel<T>, dim3((unsigned)cdiv(dI, 256), dJ, dK), dim3(256), 0, 0, a.cview(), b.cview(), dI, dJ, dK, d);
    unsigned long long h = 0;
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    hipFree(d);
    return h;
}

// ---------------------------------------------------------------------------------------------
__global__ void dpp_test_kernel(int* out_prev, int* out_next, double* dprev) {
    const int l = threadIdx.x;
    out_prev[l] = lane_from_prev(l + 100);
    out_next[l] = lane_from_next(l + 100);
    dprev[l] = lane_shift<double, true>((double)l + 0.5);
}

static bool section_dpp() {
    int *p, *n;
    double* d;
    CK(hipMalloc(&p, 64 * 4));
    CK(hipMalloc(&n, 64 * 4));
    CK(hipMalloc(&d, 64 * 8));
    hipLaunchKernelGGL(dpp_test_kernel, dim3(1), dim3(64), 0, 0, p, n, d);
    int hp[64], hn[64];
    double hd[64];
    CK(hipMemcpy(hp, p, 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hn, n, 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int l = 1; l < 64; ++l) ok &= (hp[l] == l - 1 + 100) && (hd[l] == (double)(l - 1) + 0.5);
    for (int l = 0; l < 63; ++l) ok &= (hn[l] == l + 1 + 100);
    printf("dpp        wave_shr/wave_shl self-test: %s (lane0 prev=%d, lane63 next=%d)\n", ok ? "OK" : "FAILED", hp[0], hn[63]);
    hipFree(p);
    hipFree(n);
    hipFree(d);
    return ok;
}

// ---------------------------------------------------------------------------------------------
template <int UNROLL, bool NT>
static void copy_variant(const u32x4* src, u32x4* dst, size_t nvec, unsigned blocks_per_cu) {
    size_t blocks = (nvec + 256 * UNROLL - 1) / (256 * UNROLL);
    char cfg[96];
    if (blocks_per_cu) {
        blocks = std::min<size_t>(blocks, 256ull * blocks_per_cu);
    }
    snprintf(cfg, sizeof cfg, "16B/lane unroll=%d nt=%d blocks=%zu", UNROLL, (int)NT, blocks);
    const double ms = time_ms([&](int) { hipLaunchKernelGGL((stream_copy_kernel<UNROLL, NT>), dim3((unsigned)blocks), dim3(256), 0, 0, src, dst, nvec); }, 10);
    report("copy", cfg, ms, (double)nvec, 32.0);
}

// copynt: is the streaming-copy ceiling itself a matter of cache hints?  16 bytes per lane, one vector per thread (the fastest form of
// section_copy), nontemporal store in both; the load plain or nontemporal.  A-B-A on the same two 1 GiB buffers.
template <bool NTL>
__global__ void __launch_bounds__(256) copy_ntl_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    u32x4 v;
    if constexpr (NTL) v = __builtin_nontemporal_load(&src[i]);
    else v = src[i];
    __builtin_nontemporal_store(v, &dst[i]);
}

static void section_copynt() {
    const size_t nbytes = 1ull << 30, nvec = nbytes / 16;
    char* buf[6];
    for (int n = 0; n < 6; ++n) {
        CK(hipMalloc(&buf[n], nbytes));
        CK(hipMemset(buf[n], n + 1, nbytes));
    }
    for (int pair = 0; pair < 3; ++pair) {
        const u32x4* s = reinterpret_cast<const u32x4*>(buf[0]);
        u32x4* d = reinterpret_cast<u32x4*>(buf[1 + 2 * pair]);
        for (int rep = 0; rep < 3; ++rep) {
            char cfg[64];
            double ms = time_ms([&](int) { hipLaunchKernelGGL((copy_ntl_kernel<false>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, s, d, nvec); }, 10);
            snprintf(cfg, sizeof cfg, "buffers 0 -> %d plain load, nt store", 1 + 2 * pair);
            report("copy", cfg, ms, (double)nvec, 32.0);
            ms = time_ms([&](int) { hipLaunchKernelGGL((copy_ntl_kernel<true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, s, d, nvec); }, 10);
            snprintf(cfg, sizeof cfg, "buffers 0 -> %d nt load, nt store", 1 + 2 * pair);
            report("copy", cfg, ms, (double)nvec, 32.0);
        }
    }
    for (int n = 0; n < 6; ++n) hipFree(buf[n]);
}

static void section_copy() {
    const size_t nbytes = 1ull << 30;
    char *a, *b;
    CK(hipMalloc(&a, nbytes));
    CK(hipMalloc(&b, nbytes));
    CK(hipMemset(a, 1, nbytes));
    CK(hipMemset(b, 2, nbytes));
    const size_t nvec = nbytes / 16;
    const u32x4* s = reinterpret_cast<const u32x4*>(a);
    u32x4* d = reinterpret_cast<u32x4*>(b);
    copy_variant<1, false>(s, d, nvec, 0);
    copy_variant<1, true>(s, d, nvec, 0);
    copy_variant<2, true>(s, d, nvec, 0);
    copy_variant<4, false>(s, d, nvec, 0);
    copy_variant<4, true>(s, d, nvec, 0);
    copy_variant<8, true>(s, d, nvec, 0);
    copy_variant<4, true>(s, d, nvec, 8);
    copy_variant<4, true>(s, d, nvec, 16);
    copy_variant<8, true>(s, d, nvec, 8);
    copy_variant<4, false>(s, d, nvec, 8);
    {
        const double ms = time_ms([&](int) { CK(hipMemcpyAsync(b, a, nbytes, hipMemcpyDeviceToDevice, 0)); }, 10);
        report("copy", "hipMemcpyAsync D2D", ms, (double)nvec, 32.0);
    }
    hipFree(a);
    hipFree(b);
}

// ---------------------------------------------------------------------------------------------
// Streaming ceilings per read : write mix.  A copy moves 1 : 1; horizontal diffusion with a coefficient field moves 2 : 1,
// the tridiagonal solve 4 : 3, a Laplacian 1 : 1.  What fraction of the nominal 8 TB/s a kernel of a given mix can reach is
// the yardstick its roofline fraction has to be read against (profiles/r3_microbench_rw_mix.log).
template <int NR, int NW, bool NTL = false>
__global__ void __launch_bounds__(256)
mix_kernel(const u32x4* __restrict__ r0, const u32x4* __restrict__ r1, const u32x4* __restrict__ r2, const u32x4* __restrict__ r3,
           u32x4* __restrict__ w0, u32x4* __restrict__ w1, u32x4* __restrict__ w2, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    u32x4 v = {1u, 2u, 3u, 4u};
    auto ld = [](const u32x4* p) { if constexpr (NTL) return __builtin_nontemporal_load(p); else return *p; };
    if constexpr (NR > 0) v ^= ld(&r0[i]);
    if constexpr (NR > 1) v ^= ld(&r1[i]);
    if constexpr (NR > 2) v ^= ld(&r2[i]);
    if constexpr (NR > 3) v ^= ld(&r3[i]);
    if constexpr (NW > 0) __builtin_nontemporal_store(v, &w0[i]);
    if constexpr (NW > 1) __builtin_nontemporal_store(v, &w1[i]);
    if constexpr (NW > 2) __builtin_nontemporal_store(v, &w2[i]);
    if constexpr (NW == 0)
        if (v.x == 0x12345678u && v.y == 0x9abcdef0u) w0[i] = v;  // never true: keeps the loads alive
}

template <int NR, int NW, bool NTL = false>
static void mix_variant(char* const* buf, size_t nbytes) {
    const size_t nvec = nbytes / 16;
    auto R = [&](int n) { return reinterpret_cast<const u32x4*>(buf[n]); };
    auto Wr = [&](int n) { return reinterpret_cast<u32x4*>(buf[4 + n]); };
    const double ms = time_ms([&](int) {
        hipLaunchKernelGGL((mix_kernel<NR, NW, NTL>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, R(0), R(1), R(2), R(3), Wr(0), Wr(1),
                           Wr(2), nvec);
    }, 10);
    char cfg[64];
    snprintf(cfg, sizeof cfg, "%d arrays read : %d written, 512 MiB each%s", NR, NW, NTL ? ", nt loads" : "");
    report("rw_mix", cfg, ms, (double)nvec, 16.0 * (NR + NW));
}

static void section_mix() {
    const size_t nbytes = 512ull << 20;
    char* buf[7];
    for (int n = 0; n < 7; ++n) {
        CK(hipMalloc(&buf[n], nbytes));
        CK(hipMemset(buf[n], n + 1, nbytes));
    }
    for (int rep = 0; rep < 2; ++rep) {
        mix_variant<1, 0>(buf, nbytes);
        mix_variant<2, 0>(buf, nbytes);
        mix_variant<0, 1>(buf, nbytes);
        mix_variant<1, 1>(buf, nbytes);
        mix_variant<2, 1>(buf, nbytes);
        mix_variant<3, 1>(buf, nbytes);
        mix_variant<4, 1>(buf, nbytes);
        mix_variant<4, 3>(buf, nbytes);
        mix_variant<1, 2>(buf, nbytes);
        // (round 5) the same mixes with nontemporal LOADS: the ceilings a kernel of read-once streams is measured against
        mix_variant<1, 0, true>(buf, nbytes);
        mix_variant<1, 1, true>(buf, nbytes);
        mix_variant<2, 1, true>(buf, nbytes);
        mix_variant<3, 1, true>(buf, nbytes);
        mix_variant<4, 1, true>(buf, nbytes);
        mix_variant<4, 3, true>(buf, nbytes);
    }
    for (int n = 0; n < 7; ++n) hipFree(buf[n]);
    if (getenv("MB_MIX_ONLY")) return;
    // the same 2 : 1 and 4 : 3 mixes with the arrays placed at different relative offsets inside ONE allocation: does the
    // ceiling depend on where the streams sit relative to each other (as the tridiagonal solve's five fields do)?
    {
        const size_t span = nbytes + (8ull << 20);
        char* big;
        CK(hipMalloc(&big, 7 * span + (8ull << 20)));
        CK(hipMemset(big, 1, 7 * span));
        char* base = (char*)(((uintptr_t)big + (4ull << 20) - 1) / (4ull << 20) * (4ull << 20));
        for (size_t step_kib : {0ull, 4ull, 64ull, 256ull, 1024ull, 1536ull, 2048ull, 3072ull}) {
            char* at[7];
            for (int n = 0; n < 7; ++n) at[n] = base + n * span + n * (step_kib << 10);
            // read arrays = slots 0..3, written = 4..6 (mix_variant's convention)
            for (int rep = 0; rep < 2; ++rep) {
                const size_t nvec = nbytes / 16;
                auto R = [&](int n) { return reinterpret_cast<const u32x4*>(at[n]); };
                auto Wr = [&](int n) { return reinterpret_cast<u32x4*>(at[4 + n]); };
                char cfg[96];
                double ms = time_ms([&](int) {
                    hipLaunchKernelGGL((mix_kernel<2, 1>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, R(0), R(1), R(2), R(3), Wr(0),
                                       Wr(1), Wr(2), nvec);
                }, 10);
                snprintf(cfg, sizeof cfg, "2 read : 1 written, arrays %zu KiB apart (mod 4 MiB)", step_kib);
                report("rw_place", cfg, ms, (double)nvec, 48.0);
                ms = time_ms([&](int) {
                    hipLaunchKernelGGL((mix_kernel<4, 3>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, R(0), R(1), R(2), R(3), Wr(0),
                                       Wr(1), Wr(2), nvec);
                }, 10);
                snprintf(cfg, sizeof cfg, "4 read : 3 written, arrays %zu KiB apart (mod 4 MiB)", step_kib);
                report("rw_place", cfg, ms, (double)nvec, 112.0);
            }
        }
        hipFree(big);
    }
}

// ---------------------------------------------------------------------------------------------
template <int LJ, int BLOCK, int XCD, int NTL>
static void lap_ntl_variant(const DevField<double>& in, DevField<double>& out, int dI, int dJ, int dK, const char* tag) {
    constexpr int VEC = 2;
    const unsigned tx = (unsigned)cdiv(dI, BLOCK * VEC), ty = (unsigned)cdiv(dJ, LJ);
    const unsigned n = tx * ty * dK;
    char cfg[96];
    snprintf(cfg, sizeof cfg, "%s strip LJ=%d block=%d xcd=%d nt-loads=%d", tag, LJ, BLOCK, (int)XCD, NTL);
    const double ms = time_ms([&](int) {
        hipLaunchKernelGGL((lap5_strip_kernel<double, double, 0, VEC, LJ, BLOCK, XCD, NTL>), dim3(n), dim3(BLOCK), 0, 0,
                           in.cview(), out.view(), dI, dJ, tx, ty);
    }, 20);
    report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
}

// lapnt: nontemporal loads in the Laplacian again (round 1: -4 % selective, -8 % all rows; re-measured in round 5 after the column
// kernels gained from them).  NTL 1 = only the rows no other strip reads, 2 = every row.
static void section_lapnt() {
    const int dI = 512, dJ = 512, dK = 512;
    DevField<double> in(dI, dJ, dK, 1, 1), out(dI, dJ, dK, 1, 1);
    fill(in, 5, -1.0, 1.0);
    for (int rep = 0; rep < 3; ++rep) {
        lap_ntl_variant<8, 256, 4, 0>(in, out, dI, dJ, dK, "512^3");
        lap_ntl_variant<8, 256, 4, 1>(in, out, dI, dJ, dK, "512^3");
        lap_ntl_variant<8, 256, 4, 2>(in, out, dI, dJ, dK, "512^3");
        lap_ntl_variant<16, 256, 4, 1>(in, out, dI, dJ, dK, "512^3");
        lap_ntl_variant<16, 256, 4, 0>(in, out, dI, dJ, dK, "512^3");
        if (getenv("MB_LAP_XCD")) {  // longer XCD runs: fewer strip boundaries between XCDs (the 4 % of halo re-fetch)
            lap_ntl_variant<8, 256, 8, 0>(in, out, dI, dJ, dK, "512^3");
            lap_ntl_variant<8, 256, 16, 0>(in, out, dI, dJ, dK, "512^3");
            lap_ntl_variant<8, 256, 32, 0>(in, out, dI, dJ, dK, "512^3");
            lap_ntl_variant<8, 256, 64, 0>(in, out, dI, dJ, dK, "512^3");
            lap_ntl_variant<8, 256, 2, 0>(in, out, dI, dJ, dK, "512^3");
            lap_ntl_variant<10, 256, 4, 0>(in, out, dI, dJ, dK, "512^3");
            lap_ntl_variant<12, 256, 4, 0>(in, out, dI, dJ, dK, "512^3");
        }
    }
}

// lap128: BASELINE configs[1] (512 x 512 x 128: a 90 us launch whose tail is a visible share) with four rotating pairs, workgroup shapes
template <int LJ, int BLOCK, int XCD>
static void lap128_variant(std::vector<DevField<double>*>& in, std::vector<DevField<double>*>& out, int dI, int dJ, int dK) {
    constexpr int VEC = 2;
    const unsigned tx = (unsigned)cdiv(dI, BLOCK * VEC), ty = (unsigned)cdiv(dJ, LJ);
    const unsigned n = tx * ty * dK;
    char cfg[96];
    snprintf(cfg, sizeof cfg, "512x512x128 x 4 pairs, strip LJ=%d block=%d xcd=%d", LJ, BLOCK, XCD);
    const double ms = time_ms([&](int i) {
        const int p = i & 3;
        hipLaunchKernelGGL((lap5_strip_kernel<double, double, 0, VEC, LJ, BLOCK, XCD>), dim3(n), dim3(BLOCK), 0, 0, in[p]->cview(), out[p]->view(), dI, dJ, tx, ty);
    }, 80);
    report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
}

static void section_lap128() {
    const int dI = 512, dJ = 512, dK = 128;
    std::vector<DevField<double>*> in, out;
    for (int p = 0; p < 4; ++p) {
        in.push_back(new DevField<double>(dI, dJ, dK, 1, 1));
        out.push_back(new DevField<double>(dI, dJ, dK, 1, 1));
        fill(*in.back(), 5 + p, -1.0, 1.0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        lap128_variant<8, 256, 4>(in, out, dI, dJ, dK);
        lap128_variant<8, 128, 4>(in, out, dI, dJ, dK);
        lap128_variant<8, 64, 4>(in, out, dI, dJ, dK);
        lap128_variant<8, 128, 8>(in, out, dI, dJ, dK);
        lap128_variant<4, 256, 4>(in, out, dI, dJ, dK);
        lap128_variant<6, 256, 4>(in, out, dI, dJ, dK);
        lap128_variant<8, 256, 2>(in, out, dI, dJ, dK);
    }
    for (auto* f : in) delete f;
    for (auto* f : out) delete f;
}

template <int LJ, int BLOCK, int XCD = 0>
static void lap_variant(const DevField<double>& in, DevField<double>& out, int dI, int dJ, int dK, const char* tag) {
    constexpr int VEC = 2;
    const unsigned tx = (unsigned)cdiv(dI, BLOCK * VEC), ty = (unsigned)cdiv(dJ, LJ);
    const unsigned n = tx * ty * dK;
    char cfg[96];
    snprintf(cfg, sizeof cfg, "%s strip LJ=%d block=%d xcd=%d", tag, LJ, BLOCK, (int)XCD);
    const double ms = time_ms([&](int) {
        hipLaunchKernelGGL((lap5_strip_kernel<double, double, 0, VEC, LJ, BLOCK, XCD>), dim3(n), dim3(BLOCK), 0, 0,
                           in.cview(), out.view(), dI, dJ, tx, ty);
    }, 20);
    report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
}

static void lap_suite(int dI, int dJ, int dK, int64_t extra_pitch, const char* tag) {
    DevField<double> in(dI, dJ, dK, 1, 1, 32, extra_pitch), out(dI, dJ, dK, 1, 1, 32, extra_pitch), ref(dI, dJ, dK, 1, 1, 32, extra_pitch);
    fill(in, 1337, -1.0, 1.0);
    CK(hipMemset(out.raw, 0, out.bytes));
    CK(hipMemset(ref.raw, 0, ref.bytes));
    {
        dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
        const double ms = time_ms([&](int) {
            hipLaunchKernelGGL((lap5_generic_kernel<double, double, 0>), grid, dim3(256), 0, 0, in.cview(), ref.view(), dI, dJ, dK);
        }, 5, 1);
        char cfg[64];
        snprintf(cfg, sizeof cfg, "%s generic (one thread per point)", tag);
        report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
    }
    for (int rep = 0; rep < 2; ++rep) {
        lap_variant<4, 256>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 256>(in, out, dI, dJ, dK, tag);
        if (rep == 0) printf("           check vs generic: %llu mismatches\n", count_diff(out, ref, dI, dJ, dK));
        lap_variant<8, 256, -1>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 256, 2>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 256, 4>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 256, 8>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 256, 16>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 256, 64>(in, out, dI, dJ, dK, tag);
        lap_variant<4, 256, 4>(in, out, dI, dJ, dK, tag);
        lap_variant<4, 256, 16>(in, out, dI, dJ, dK, tag);
        lap_ntl_variant<8, 256, 4, 0>(in, out, dI, dJ, dK, tag);
        lap_ntl_variant<8, 256, 4, 1>(in, out, dI, dJ, dK, tag);
        lap_ntl_variant<8, 256, 4, 2>(in, out, dI, dJ, dK, tag);
        if (rep == 0) printf("           check nt-loads vs generic: %llu mismatches\n", count_diff(out, ref, dI, dJ, dK));
        lap_ntl_variant<8, 256, 8, 1>(in, out, dI, dJ, dK, tag);
        lap_ntl_variant<8, 256, 0, 1>(in, out, dI, dJ, dK, tag);
        lap_variant<12, 256>(in, out, dI, dJ, dK, tag);
        lap_variant<16, 256>(in, out, dI, dJ, dK, tag);
        lap_variant<16, 256, 4>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 128>(in, out, dI, dJ, dK, tag);
        lap_variant<16, 128>(in, out, dI, dJ, dK, tag);
        lap_variant<8, 64>(in, out, dI, dJ, dK, tag);
        const int64_t d[3] = {dI, dJ, dK};
        const double ms = time_ms([&](int) { lap5_launch_variant<double, double, 0>(in.cview(), out.view(), d, 0); }, 20);
        char cfg[64];
        snprintf(cfg, sizeof cfg, "%s library default", tag);
        report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
        if (rep == 0) printf("           check vs generic: %llu mismatches\n", count_diff(out, ref, dI, dJ, dK));
    }
}

// lapshare (round 6): the Laplacian strip kernel with the J halo rows of neighbouring waves exchanged through LDS (experiments/lap5_share.hip.h)
template <int LJ, int NWI, int NWJ, int XCDG>
static void lapshare_variant(const DevField<double>& in, DevField<double>& out, const DevField<double>& ref, int dI, int dJ, int dK, const char* tag) {
    char cfg[128];
    snprintf(cfg, sizeof cfg, "%s shared halo rows LJ=%d waves %d x %d xcd=%d", tag, LJ, NWI, NWJ, XCDG);
    CK(hipMemset(out.raw, 0, out.bytes));
    const double ms = time_ms([&](int) { lap5_share_launch<double, double, 0, 2, LJ, NWI, NWJ, XCDG>(in.cview(), out.view(), dI, dJ, dK, 0); }, 30);
    CK(hipGetLastError());
    report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
    const unsigned long long bad = count_diff(out, ref, dI, dJ, dK);
    if (bad) printf("           MISMATCHES vs the one-thread-per-point kernel: %llu\n", bad);
}

static void lapshare_suite(int dI, int dJ, int dK, const char* tag) {
    DevField<double> in(dI, dJ, dK, 1, 1, 16, 0), out(dI, dJ, dK, 1, 1, 16, 0), ref(dI, dJ, dK, 1, 1, 16, 0);  // rows on 128-byte boundaries: the storage preset
    fill(in, 1337, -1.0, 1.0);
    CK(hipMemset(ref.raw, 0, ref.bytes));
    {
        dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
        hipLaunchKernelGGL((lap5_generic_kernel<double, double, 0>), grid, dim3(256), 0, 0, in.cview(), ref.view(), dI, dJ, dK);
        CK(hipDeviceSynchronize());
    }
    const int64_t d[3] = {dI, dJ, dK};
    for (int rep = 0; rep < 3; ++rep) {
        {
            const double ms = time_ms([&](int) { lap5_launch_variant<double, double, 0>(in.cview(), out.view(), d, 0); }, 30);
            char cfg[96];
            snprintf(cfg, sizeof cfg, "%s library (strip LJ=8 block=256 xcd=4)", tag);
            report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
        }
        lapshare_variant<8, 4, 1, 4>(in, out, ref, dI, dJ, dK, tag);  // the library's tile with the exchange machinery but nothing to exchange
        lapshare_variant<8, 4, 2, 2>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 4, 2, 4>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 4, 2, 1>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 4, 4, 1>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 4, 4, 2>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<4, 4, 2, 4>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<4, 4, 4, 2>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<4, 4, 4, 4>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<6, 4, 4, 2>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 2, 2, 4>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 2, 4, 2>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 1, 4, 4>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<8, 1, 8, 2>(in, out, ref, dI, dJ, dK, tag);
        lapshare_variant<16, 4, 2, 2>(in, out, ref, dI, dJ, dK, tag);
    }
}

static void section_lapshare() {
    lapshare_suite(512, 512, 512, "512^3");
    lapshare_suite(512, 512, 128, "512x512x128");
    lapshare_suite(334, 131, 7, "334x131x7");
}

// lappersist (round 6): one workgroup per tile (the library) against a persistent grid looping over the tiles; N rotating (in, out) pairs
static void lappersist_suite(int dI, int dJ, int dK, int npairs, const char* tag) {
    std::vector<DevField<double>*> in, out;
    for (int p = 0; p < npairs; ++p) {
        in.push_back(new DevField<double>(dI, dJ, dK, 1, 1, 16, 0));
        out.push_back(new DevField<double>(dI, dJ, dK, 1, 1, 16, 0));
        fill(*in.back(), 1337 + p, -1.0, 1.0);
        CK(hipMemset(out.back()->raw, 0, out.back()->bytes));
    }
    DevField<double> ref(dI, dJ, dK, 1, 1, 16, 0);
    CK(hipMemset(ref.raw, 0, ref.bytes));
    {
        dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
        hipLaunchKernelGGL((lap5_generic_kernel<double, double, 0>), grid, dim3(256), 0, 0, in[0]->cview(), ref.view(), dI, dJ, dK);
        CK(hipDeviceSynchronize());
    }
    const int64_t d[3] = {dI, dJ, dK};
    const unsigned tx = (unsigned)cdiv(dI, 512), ty = (unsigned)cdiv(dJ, 8), n = tx * ty * (unsigned)dK;
    const int iters = 40 * npairs;
    for (int rep = 0; rep < 3; ++rep) {
        char cfg[128];
        {
            const double ms = time_ms([&](int i) { lap5_launch_variant<double, double, 0>(in[i % npairs]->cview(), out[i % npairs]->view(), d, 0); }, iters, npairs);
            snprintf(cfg, sizeof cfg, "%s library: %u workgroups, %d rotating pairs", tag, n, npairs);
            report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
        }
        for (unsigned per_cu : {2u, 4u, 6u, 8u, 12u, 16u}) {
            const unsigned grid = 256u * per_cu < n ? 256u * per_cu : n;
            const double ms = time_ms([&](int i) {
                hipLaunchKernelGGL((lap5_persistent_kernel<double, double, 0, 2, 8, 256, 4>), dim3(grid), dim3(256), 0, 0, in[i % npairs]->cview(),
                                   out[i % npairs]->view(), dI, dJ, tx, ty, n);
            }, iters, npairs);
            snprintf(cfg, sizeof cfg, "%s persistent grid of %u workgroups (%u per CU)", tag, grid, per_cu);
            report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
            if (rep == 0) {
                const unsigned long long bad = count_diff(*out[0], ref, dI, dJ, dK);
                if (bad) printf("           MISMATCHES vs the one-thread-per-point kernel: %llu\n", bad);
            }
        }
    }
    for (auto* f : in) delete f;
    for (auto* f : out) delete f;
}

static void section_lappersist() {
    lappersist_suite(512, 512, 128, 4, "512x512x128");
    lappersist_suite(512, 512, 512, 2, "512^3");
}

static void section_lap() {
    lap_suite(512, 512, 512, 0, "512^3");
    lap_suite(512, 512, 128, 0, "512x512x128");
    lap_suite(128, 256, 512, 0, "128x256x512");
    lap_suite(256, 256, 512, 0, "256x256x512");
}

// ---------------------------------------------------------------------------------------------
template <typename T, typename W, int VEC, int LJ, int PF, int XCDG = 0>
static void hdiff_variant(const DevField<T>& in, DevField<T>& out, const DevField<T>& cf, int dI, int dJ, int dK, const char* tag,
                          double bpl) {
    constexpr int H = (VEC >= 2) ? 1 : 2;
    const unsigned waves_i = (unsigned)cdiv(dI, (64 - 2 * H) * VEC), tiles_j = (unsigned)cdiv(dJ, LJ);
    const unsigned groups_j = (unsigned)cdiv(tiles_j, 4);
    const unsigned nb = waves_i * groups_j * dK;
    char cfg[96];
    snprintf(cfg, sizeof cfg, "%s jmarch VEC=%d LJ=%d PF=%d xcd=%d", tag, VEC, LJ, PF, XCDG);
    const double ms = time_ms([&](int) {
        hipLaunchKernelGGL((hdiff_jmarch_kernel<T, W, W, true, true, VEC, LJ, PF, XCDG>), dim3(nb), dim3(256), 0, 0,
                           in.cview(), out.view(), cf.cview(), (W)0, dI, dJ, waves_i, tiles_j, groups_j, 0);
    }, 20);
    report(sizeof(T) == 4 ? "hdiff_f32" : "hdiff_f64", cfg, ms, (double)dI * dJ * dK, bpl);
}

template <typename T>
static void hdiff_suite(int dI, int dJ, int dK, const char* tag) {
    using W = double;
    constexpr int VMAX = 16 / sizeof(T);
    const double bpl = 3.0 * sizeof(T);
    DevField<T> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), ref(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
    fill(in, 2024, 1.0, 9.0);
    fill(cf, 7, 0.0, 0.05);
    CK(hipMemset(out.raw, 0, out.bytes));
    CK(hipMemset(ref.raw, 0, ref.bytes));
    {
        dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
        const double ms = time_ms([&](int) {
            hipLaunchKernelGGL((hdiff_generic_kernel<T, W, W, true, true>), grid, dim3(256), 0, 0, in.cview(), ref.view(), cf.cview(), (W)0, dI, dJ, dK);
        }, 5, 1);
        char cfg[64];
        snprintf(cfg, sizeof cfg, "%s generic (one thread per point)", tag);
        report(sizeof(T) == 4 ? "hdiff_f32" : "hdiff_f64", cfg, ms, (double)dI * dJ * dK, bpl);
    }
    hdiff_variant<T, W, VMAX, 12, 6, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
    printf("           check vs generic: %llu mismatches\n", count_diff(out, ref, dI, dJ, dK));
    CK(hipMemset(out.raw, 0, out.bytes));
    hdiff_variant<T, W, 1, 8, 8, 0>(in, out, cf, dI, dJ, dK, tag, bpl);
    printf("           check vs generic: %llu mismatches\n", count_diff(out, ref, dI, dJ, dK));
    for (int rep = 0; rep < 2; ++rep) {
        hdiff_variant<T, W, VMAX, 8, 8, 0>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 8, 8, 2>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 8, 8, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 8, 8, 8>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 12, 6, 0>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 12, 6, 2>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 12, 6, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 12, 12, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 16, 8, 0>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 16, 8, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 16, 16, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, VMAX, 6, 6, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
        hdiff_variant<T, W, 2, 8, 8, 4>(in, out, cf, dI, dJ, dK, tag, bpl);
    }
    {
        const int64_t d[3] = {dI, dJ, dK};
        const double ms = time_ms([&](int) { hdiff_launch<T, W, W, true, true>(in.cview(), out.view(), cf.cview(), (W)0, d, 0); }, 20);
        char cfg[64];
        snprintf(cfg, sizeof cfg, "%s library default", tag);
        report(sizeof(T) == 4 ? "hdiff_f32" : "hdiff_f64", cfg, ms, (double)dI * dJ * dK, bpl);
        printf("           check vs generic: %llu mismatches\n", count_diff(out, ref, dI, dJ, dK));
    }
}

template <typename T, typename W, int VEC, int LJ, int PF, int XCDG>
static void hdiff_variant_w(const DevField<T>& in, DevField<T>& out, const DevField<T>& cf, int dI, int dJ, int dK, const char* tag) {
    constexpr int H = (VEC >= 2) ? 1 : 2;
    const unsigned waves_i = (unsigned)cdiv(dI, (64 - 2 * H) * VEC), tiles_j = (unsigned)cdiv(dJ, LJ);
    const unsigned groups_j = (unsigned)cdiv(tiles_j, 4);
    const unsigned nb = waves_i * (XCDG < 0 ? (unsigned)cdiv(groups_j, 8) * 8u : groups_j) * dK;
    char cfg[96];
    snprintf(cfg, sizeof cfg, "%s %s-internal VEC=%d LJ=%d PF=%d xcd=%d", tag, sizeof(W) == 4 ? "f32" : "f64", VEC, LJ, PF, XCDG);
    const double ms = time_ms([&](int) {
        hipLaunchKernelGGL((hdiff_jmarch_kernel<T, W, W, true, true, VEC, LJ, PF, XCDG>), dim3(nb), dim3(256), 0, 0,
                           in.cview(), out.view(), cf.cview(), (W)0, dI, dJ, waves_i, tiles_j, groups_j, 0);
    }, 20);
    report(sizeof(T) == 4 ? "hdiff_f32" : "hdiff_f64", cfg, ms, (double)dI * dJ * dK, 3.0 * sizeof(T));
}

// f32 fields with f32 literals (literal_float_precision=32) next to the default f64 internals, and short queues
static void section_hdiff2() {
    {
        const int dI = 1024, dJ = 1024, dK = 80;
        DevField<float> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        for (int rep = 0; rep < 2; ++rep) {
            hdiff_variant_w<float, double, 4, 6, 6, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 8, 8, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 8, 6, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 8, 4, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 4, 4, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 8, 8, 2>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, float, 4, 6, 6, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, float, 4, 8, 8, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, float, 4, 16, 8, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, float, 4, 12, 12, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
        }
    }
    {
        const int dI = 512, dJ = 1024, dK = 80;
        DevField<double> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        for (int rep = 0; rep < 2; ++rep) {
            hdiff_variant_w<double, double, 2, 8, 8, 4>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 8, 8, 2>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 8, 6, 4>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 8, 4, 4>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 4, 4, 4>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 16, 16, 4>(in, out, cf, dI, dJ, dK, "512x1024x80");
        }
    }
}

// Round 5: the float32 / float64-internal kernel is VALU co-limited.  Value-identical variants of the
// instruction mix (OPT), longer strips behind a short prefetch window, and the "team" mapping (four waves on four adjacent I
// strips of the same rows); every variant is compared with the one-thread-per-point kernel bit for bit.
template <typename T, typename W, int VEC, int LJ, int PF, int XCDG, int OPT, bool TEAM>
static void hdiff_variant_o(const DevField<T>& in, DevField<T>& out, const DevField<T>& ref, const DevField<T>& cf, int dI, int dJ, int dK,
                            const char* tag) {
    constexpr int H = (VEC >= 2) ? 1 : 2;
    const unsigned waves_i = (unsigned)cdiv(dI, (64 - 2 * H) * VEC), tiles_j = (unsigned)cdiv(dJ, LJ);
    const unsigned groups_j = TEAM ? tiles_j : (unsigned)cdiv(tiles_j, 4);
    const unsigned nb = (TEAM ? (unsigned)cdiv(waves_i, 4) : waves_i) * ((XCDG < 0 && !TEAM) ? (unsigned)cdiv(groups_j, 8) * 8u : groups_j) * dK;
    char cfg[128];
    snprintf(cfg, sizeof cfg, "%s %s-internal VEC=%d LJ=%d PF=%d xcd=%d opt=%d%s", tag, sizeof(W) == 4 ? "f32" : "f64", VEC, LJ, PF, XCDG,
             OPT, TEAM ? " TEAM" : "");
    CK(hipMemset(out.raw, 0, out.bytes));
    const double ms = time_ms([&](int) {
        hipLaunchKernelGGL((hdiff_jmarch_kernel<T, W, W, true, true, VEC, LJ, PF, XCDG, OPT, TEAM>), dim3(nb), dim3(256), 0, 0,
                           in.cview(), out.view(), cf.cview(), (W)0, dI, dJ, waves_i, tiles_j, groups_j, 0);
    }, 100);
    report(sizeof(T) == 4 ? "hdiff_f32" : "hdiff_f64", cfg, ms, (double)dI * dJ * dK, 3.0 * sizeof(T));
    const unsigned long long bad = count_diff(out, ref, dI, dJ, dK);
    if (bad) printf("           MISMATCHES vs the one-thread-per-point kernel: %llu\n", bad);
}

static void section_hdiff3() {
    {
        const int dI = 1024, dJ = 1024, dK = 80;
        DevField<float> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), ref(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        {
            dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
            hipLaunchKernelGGL((hdiff_generic_kernel<float, double, double, true, true>), grid, dim3(256), 0, 0, in.cview(), ref.view(), cf.cview(),
                               0.0, dI, dJ, dK);
            CK(hipDeviceSynchronize());
        }
        for (int rep = 0; rep < 2; ++rep) {
#define V(LJ, PF, X, O, TM) hdiff_variant_o<float, double, 4, LJ, PF, X, O, TM>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80")
            V(6, 6, 4, 0, false);  // the library's kernel
            V(6, 6, 4, 3, false);  // packed f32 pairs + kept wide copies
            V(8, 8, 4, 0, false);
            V(8, 4, 4, 0, false);
            V(6, 6, 4, 4, false);  // the same strips, rolled (ring queue)
            V(12, 6, 4, 4, false);
            V(12, 4, 4, 4, false);
            V(12, 3, 4, 4, false);
            V(24, 6, 4, 4, false);
            V(24, 4, 4, 4, false);
            V(12, 6, 4, 4, true);  // four waves on four adjacent I strips of the same rows
            V(24, 6, 4, 4, true);
            V(24, 4, 4, 4, true);
            V(24, 3, 4, 4, true);
            V(48, 6, 4, 4, true);
            V(48, 4, 4, 4, true);
            V(48, 6, 0, 4, true);
            V(48, 6, 2, 4, true);
            V(96, 6, 4, 4, true);
            V(48, 6, 4, 7, true);
            V(128, 4, 4, 4, true);
#undef V
        }
    }
    {
        const int dI = 512, dJ = 1024, dK = 80;
        DevField<double> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), ref(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        {
            dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
            hipLaunchKernelGGL((hdiff_generic_kernel<double, double, double, true, true>), grid, dim3(256), 0, 0, in.cview(), ref.view(), cf.cview(),
                               0.0, dI, dJ, dK);
            CK(hipDeviceSynchronize());
        }
        for (int rep = 0; rep < 2; ++rep) {
#define V(LJ, PF, X, O, TM) hdiff_variant_o<double, double, 2, LJ, PF, X, O, TM>(in, out, ref, cf, dI, dJ, dK, "512x1024x80")
            V(8, 8, 4, 0, false);  // the library's kernel
            V(8, 4, 4, 4, false);
            V(16, 4, 4, 4, false);
            V(24, 6, 4, 4, true);
            V(48, 6, 4, 4, true);
            V(48, 4, 4, 4, true);
            V(96, 4, 4, 4, true);
#undef V
        }
    }
}

// hdiffnt: nontemporal loads in the J-march kernel (coeff is read exactly once; `in` has halo rows that other strips re-read)
static void section_hdiffnt() {
    {
        const int dI = 1024, dJ = 1024, dK = 80;
        DevField<float> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), ref(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        {
            dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
            hipLaunchKernelGGL((hdiff_generic_kernel<float, double, double, true, true>), grid, dim3(256), 0, 0, in.cview(), ref.view(), cf.cview(),
                               0.0, dI, dJ, dK);
            CK(hipDeviceSynchronize());
        }
        for (int rep = 0; rep < 3; ++rep) {
#define V(O) hdiff_variant_o<float, double, 4, 6, 6, 4, O, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80")
            V(0);
            V(8);   // nt coeff
            V(16);  // nt in
            V(24);  // both
#undef V
            // workgroup -> XCD order with the nontemporal coeff loads: runs of 2 / 8 workgroups, one contiguous J chunk per XCD (-1)
            hdiff_variant_o<float, double, 4, 6, 6, 2, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 6, 6, 8, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 6, 6, -1, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 6, 6, 0, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 4, 4, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 5, 5, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 6, 4, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 8, 6, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_o<float, double, 4, 8, 8, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "1024x1024x80");
        }
    }
    {
        const int dI = 512, dJ = 1024, dK = 80;
        DevField<double> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), ref(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        {
            dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
            hipLaunchKernelGGL((hdiff_generic_kernel<double, double, double, true, true>), grid, dim3(256), 0, 0, in.cview(), ref.view(), cf.cview(),
                               0.0, dI, dJ, dK);
            CK(hipDeviceSynchronize());
        }
        for (int rep = 0; rep < 3; ++rep) {
#define V(O) hdiff_variant_o<double, double, 2, 8, 8, 4, O, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80")
            V(0);
            V(8);
            V(16);
            V(24);
#undef V
            // shorter strips with the nontemporal coeff loads (the GENERATED kernel, 4 rows per lane, is 2-4 % ahead of 8 / 8 in bench.py)
            hdiff_variant_o<double, double, 2, 6, 6, 2, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 6, 6, 8, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 6, 6, -1, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 4, 4, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 6, 6, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 6, 4, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 8, 6, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 10, 8, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_o<double, double, 2, 12, 8, 4, 8, false>(in, out, ref, cf, dI, dJ, dK, "512x1024x80");
        }
    }
}

// workgroup -> XCD mappings of the J-march kernel: runs of G workgroups (0, 2, 4, 8) and contiguous chunks per column (-1)
static void section_hdiffxcd() {
    {
        const int dI = 1024, dJ = 1024, dK = 80;
        DevField<float> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        for (int rep = 0; rep < 2; ++rep) {
            hdiff_variant_w<float, double, 4, 6, 6, 0>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 6, 6, 2>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 6, 6, 4>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 6, 6, 8>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 6, 6, -1>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 8, 8, -1>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 8, 4, -1>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, double, 4, 4, 4, -1>(in, out, cf, dI, dJ, dK, "1024x1024x80");
            hdiff_variant_w<float, float, 4, 6, 6, -1>(in, out, cf, dI, dJ, dK, "1024x1024x80");
        }
    }
    {
        const int dI = 512, dJ = 1024, dK = 80;
        DevField<double> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
        fill(in, 2024, 1.0, 9.0);
        fill(cf, 7, 0.0, 0.05);
        for (int rep = 0; rep < 2; ++rep) {
            hdiff_variant_w<double, double, 2, 8, 8, 4>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 8, 8, -1>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 6, 6, 4>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 6, 6, -1>(in, out, cf, dI, dJ, dK, "512x1024x80");
            hdiff_variant_w<double, double, 2, 4, 4, -1>(in, out, cf, dI, dJ, dK, "512x1024x80");
        }
    }
}


// MB_HDIFFL=<substring>[,<substring>...]: keep only the hdiffl variants whose configuration line contains one of them
static bool hdiffl_selected(const char* cfg) {
    const char* only = getenv("MB_HDIFFL");
    if (!only || !*only) return true;
    std::string all(only);
    size_t a = 0;
    while (a <= all.size()) {
        size_t b = all.find(',', a);
        if (b == std::string::npos) b = all.size();
        if (b > a && strstr(cfg, all.substr(a, b - a).c_str())) return true;
        a = b + 1;
    }
    return false;
}

// hdiffl (round 6): the LDS-staged `in` block (experiments/hdiff_ldstile.hip.h) beside the library's register-only J-march, every
// variant compared bit for bit with the one-thread-per-point kernel.  MB_HDIFFL=<substring> keeps only the variants whose
// configuration line contains it (for one-kernel PMC passes); MB_HDIFFL_ITERS sets the timed launches per variant.
template <typename T, typename W, int VEC, int RW, int NW, bool GLDS>
static void hdiff_variant_l(const DevField<T>& in, DevField<T>& out, const DevField<T>& ref, const DevField<T>& cf, int dI, int dJ, int dK,
                            const char* tag, int seg_rows, int xcd_group) {
    char cfg[160];
    snprintf(cfg, sizeof cfg, "%s %s-internal LDS ring RW=%d NW=%d %s seg=%d xcd=%d", tag, sizeof(W) == 4 ? "f32" : "f64", RW, NW,
             GLDS ? "glds" : "regs", seg_rows, xcd_group);
    if (!hdiffl_selected(cfg)) return;
    const int iters = getenv("MB_HDIFFL_ITERS") ? atoi(getenv("MB_HDIFFL_ITERS")) : 100;
    CK(hipMemset(out.raw, 0, out.bytes));
    const double ms = time_ms([&](int) {
        hdiff_ldstile_launch<T, W, VEC, RW, NW, GLDS>(in.cview(), out.view(), cf.cview(), dI, dJ, dK, seg_rows, xcd_group, 0);
    }, iters);
    CK(hipGetLastError());
    report(sizeof(T) == 4 ? "hdiff_f32" : "hdiff_f64", cfg, ms, (double)dI * dJ * dK, 3.0 * sizeof(T));
    const unsigned long long bad = count_diff(out, ref, dI, dJ, dK);
    if (bad) printf("           MISMATCHES vs the one-thread-per-point kernel: %llu\n", bad);
}

template <typename T, typename W, int VEC, int LJ, int NW, int XCDG, int MINW = 4>
static void hdiff_variant_s(const DevField<T>& in, DevField<T>& out, const DevField<T>& ref, const DevField<T>& cf, int dI, int dJ, int dK,
                            const char* tag) {
    char cfg[160];
    snprintf(cfg, sizeof cfg, "%s %s-internal shared halo rows in LDS LJ=%d NW=%d minw=%d xcd=%d", tag, sizeof(W) == 4 ? "f32" : "f64", LJ, NW, MINW,
             XCDG);
    if (!hdiffl_selected(cfg)) return;
    const int iters = getenv("MB_HDIFFL_ITERS") ? atoi(getenv("MB_HDIFFL_ITERS")) : 100;
    CK(hipMemset(out.raw, 0, out.bytes));
    const int64_t d[3] = {dI, dJ, dK};
    const double ms = time_ms([&](int) {
        (void)hdiff_launch_share_shape<T, W, W, true, true, VEC, LJ, NW, XCDG, MINW>(in.cview(), out.view(), cf.cview(), (W)0, d, 0, 0);
    }, iters);
    CK(hipGetLastError());
    report(sizeof(T) == 4 ? "hdiff_f32" : "hdiff_f64", cfg, ms, (double)dI * dJ * dK, 3.0 * sizeof(T));
    const unsigned long long bad = count_diff(out, ref, dI, dJ, dK);
    if (bad) printf("           MISMATCHES vs the one-thread-per-point kernel: %llu\n", bad);
}

template <typename T, typename W, int VEC>
static void hdiffl_suite(int dI, int dJ, int dK, const char* tag) {
    DevField<T> in(dI, dJ, dK, 2, 2), out(dI, dJ, dK, 2, 2), ref(dI, dJ, dK, 2, 2), cf(dI, dJ, dK, 2, 2);
    fill(in, 2024, 1.0, 9.0);
    fill(cf, 7, 0.0, 0.05);
    {
        dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4), (unsigned)dK);
        hipLaunchKernelGGL((hdiff_generic_kernel<T, W, W, true, true>), grid, dim3(256), 0, 0, in.cview(), ref.view(), cf.cview(), (W)0, dI, dJ, dK);
        CK(hipDeviceSynchronize());
    }
    const int reps = getenv("MB_HDIFFL_REPS") ? atoi(getenv("MB_HDIFFL_REPS")) : 2;
    for (int rep = 0; rep < reps; ++rep) {
        if (hdiffl_selected("register J-march"))
            hdiff_variant_o<T, W, VEC, 6, 6, 4, HD_OPT_NT_COEFF, false>(in, out, ref, cf, dI, dJ, dK, tag);  // the library's kernel of rounds 1-5
#define L(RW, NW, G, SEG, X) hdiff_variant_l<T, W, VEC, RW, NW, G>(in, out, ref, cf, dI, dJ, dK, tag, SEG, X)
        L(6, 4, true, 128, 4);
        L(6, 4, false, 128, 4);
        L(4, 4, true, 128, 4);
        L(4, 4, false, 128, 4);
        L(8, 4, true, 128, 4);
        L(8, 4, true, 256, 4);
        L(4, 8, true, 256, 4);
#undef L
#define S(LJ, NW, X, MW) hdiff_variant_s<T, W, VEC, LJ, NW, X, MW>(in, out, ref, cf, dI, dJ, dK, tag)
        S(4, 4, 2, 4);  // the library's shape
        S(4, 4, 0, 4);
        S(4, 4, 1, 4);
        S(4, 4, 4, 4);
        S(4, 4, 8, 4);
        S(4, 4, 16, 4);
        S(4, 2, 4, 4);
        S(4, 2, 8, 4);
        S(4, 3, 4, 4);
        S(4, 6, 2, 4);
        S(4, 8, 1, 4);
        S(5, 4, 4, 4);
        S(6, 4, 4, 3);
        S(6, 4, 4, 4);
        S(4, 16, 1, 4);
#undef S
    }
}

static void section_hdiffl() {
    hdiffl_suite<float, double, 4>(1024, 1024, 80, "1024x1024x80");
    hdiffl_suite<double, double, 2>(512, 1024, 80, "512x1024x80");
    if (getenv("MB_HDIFFL_BIG_ONLY")) return;
    // ragged domains: partial strips, a partial last chunk, a last segment shorter than a chunk
    hdiffl_suite<double, double, 2>(333, 517, 5, "333x517x5");
    hdiffl_suite<float, double, 4>(250, 131, 3, "250x131x3");
}

static void section_hdiff() {
    hdiff_suite<float>(1024, 1024, 80, "1024x1024x80");
    hdiff_suite<double>(512, 1024, 80, "512x1024x80");
}

// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int UNROLL>
static void tridiag_variant(DevField<T>& a, DevField<T>& d, DevField<T>& s, DevField<T>& r, DevField<T>& o, int dI, int dJ, int dK,
                            const char* tag) {
    const unsigned ti = (unsigned)cdiv(dI, 256 * VEC);
    char cfg[96];
    snprintf(cfg, sizeof cfg, "%s VEC=%d UNROLL=%d", tag, VEC, UNROLL);
    const double ms = time_ms([&](int) {
        hipLaunchKernelGGL((tridiag_kernel<T, VEC, UNROLL>), dim3(ti * dJ), dim3(256), 0, 0, a.cview(), d.cview(), s.view(), r.view(), o.view(), dI, dJ, dK, ti);
    }, 5, 1);
    report("tridiag64", cfg, ms, (double)dI * dJ * dK, 56.0);
}

template <int RL, int LL, int U, bool PIPE = false, int WPB = 1, int MAP = 0, int NTL = 0>
static void tridiag_stack_variant(DevField<double>& a, DevField<double>& d, DevField<double>& s, DevField<double>& r,
                                  DevField<double>& o, DevField<double>& s2, DevField<double>& r2, DevField<double>& o2,
                                  int dI, int dJ, int dK) {
    const unsigned ti = (unsigned)cdiv(dI, 64);
    char cfg[96];
    snprintf(cfg, sizeof cfg, "%s RL=%d LL=%d U=%d WPB=%d MAP=%d%s (mem levels %d)", PIPE ? "pipe " : "stack", RL, LL, U, WPB, MAP,
             NTL == 1 ? " nt loads (all four)" : NTL == 2 ? " nt loads (inf, diag)" : "", dK - RL - LL);
    if (dK - RL - LL < 1) return;
    auto launch = [&]() {
        if constexpr (PIPE)
            hipLaunchKernelGGL((tridiag_pipe_kernel<double, RL, LL, U, WPB, MAP, NTL>), dim3(ti * (unsigned)cdiv(dJ, WPB)), dim3(64, WPB), 0, 0, a.cview(), d.cview(), s.view(), r.view(), o.view(), dI, dJ, dK, ti);
        else
            hipLaunchKernelGGL((tridiag_stack_kernel<double, RL, LL, U>), dim3(ti * dJ), dim3(64), 0, 0, a.cview(), d.cview(), s.view(), r.view(), o.view(), dI, dJ, dK, ti);
    };
    // correctness against the two-sweep kernel from identical inputs
    fill(s, 3, -1.0, 1.0);
    fill(r, 4, -10.0, 10.0);
    fill(s2, 3, -1.0, 1.0);
    fill(r2, 4, -10.0, 10.0);
    CK(hipMemset(o.raw, 0xff, o.bytes));
    launch();
    {
        const unsigned t2 = (unsigned)cdiv(dI, 256);
        hipLaunchKernelGGL((tridiag_kernel<double, 1, 8>), dim3(t2 * dJ), dim3(256), 0, 0, a.cview(), d.cview(), s2.view(), r2.view(), o2.view(), dI, dJ, dK, t2);
    }
    CK(hipDeviceSynchronize());
    printf("           check %s: out %llu sup %llu rhs %llu mismatches\n", cfg, count_diff(o, o2, dI, dJ, dK),
           count_diff(s, s2, dI, dJ, dK), count_diff(r, r2, dI, dJ, dK));
    const double ms = time_ms([&](int) { launch(); }, 5, 1);
    report("tridiag64", cfg, ms, (double)dI * dJ * dK, 56.0);
}

// The same solve on fields whose K levels are one ROW PITCH apart (layout I, K, J from fastest to slowest) instead of one
// plane apart: a column's 160 levels then lie inside one or two 2 MiB pages.  Tests the address-translation hypothesis.
static void section_trilayout() {
    const int dI = 1024, dJ = 1024, dK = 160;
    const size_t n = (size_t)dI * dJ * dK;
    double* p[5];
    for (int f = 0; f < 5; ++f) CK(hipMalloc(&p[f], n * sizeof(double) + 4096));
    const unsigned ti = (unsigned)cdiv(dI, 64), t2 = (unsigned)cdiv(dI, 256);
    for (int rep = 0; rep < 2; ++rep)
        for (int layout = 0; layout < 2; ++layout) {
            const int64_t sj = layout == 0 ? dI : (int64_t)dI * dK, sk = layout == 0 ? (int64_t)dI * dJ : dI;
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[0], n, 1u, -1.0, 1.0);
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[1], n, 2u, 4.0, 5.0);
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[2], n, 3u, -1.0, 1.0);
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[3], n, 4u, -10.0, 10.0);
            CK(hipDeviceSynchronize());
            View<const double> a{p[0], 1, sj, sk}, d{p[1], 1, sj, sk};
            View<double> s{p[2], 1, sj, sk}, r{p[3], 1, sj, sk}, o{p[4], 1, sj, sk};
            const char* tag = layout == 0 ? "planes: K stride 8 MiB (I,J,K)" : "rows:   K stride 8 KiB (I,K,J)";
            char cfg[96];
            double ms = time_ms([&](int) {
                hipLaunchKernelGGL((tridiag_pipe_kernel<double, 80, 40, 8>), dim3(ti * dJ), dim3(64), 0, 0, a, d, s, r, o, dI, dJ, dK, ti);
            }, 5, 2);
            snprintf(cfg, sizeof cfg, "%s pipe<80,40,8>", tag);
            report("trilayout", cfg, ms, (double)n, 56.0);
            ms = time_ms([&](int) {
                hipLaunchKernelGGL((tridiag_pipe_kernel<double, 32, 40, 8>), dim3(ti * dJ), dim3(64), 0, 0, a, d, s, r, o, dI, dJ, dK, ti);
            }, 5, 2);
            snprintf(cfg, sizeof cfg, "%s pipe<32,40,8>", tag);
            report("trilayout", cfg, ms, (double)n, 56.0);
            ms = time_ms([&](int) {
                hipLaunchKernelGGL((tridiag_kernel<double, 1, 8>), dim3(t2 * dJ), dim3(256), 0, 0, a, d, s, r, o, dI, dJ, dK, t2);
            }, 5, 2);
            snprintf(cfg, sizeof cfg, "%s two-sweep<1,8>", tag);
            report("trilayout", cfg, ms, (double)n, 56.0);
        }
    for (int f = 0; f < 5; ++f) hipFree(p[f]);
}

// a few launches of each column kernel and of the Laplacian, for counter passes (rocprofv3 --pmc ...)
static void section_tripmc() {
    {
        const int dI = 1024, dJ = 1024, dK = 160;
        DevField<double> a(dI, dJ, dK, 0, 0), d(dI, dJ, dK, 0, 0), s(dI, dJ, dK, 0, 0), r(dI, dJ, dK, 0, 0), o(dI, dJ, dK, 0, 0);
        DevField<double> s2(dI, dJ, dK, 0, 0), r2(dI, dJ, dK, 0, 0), o2(dI, dJ, dK, 0, 0);
        printf("tripmc     fields at %p %p %p %p %p\n", (void*)a.data, (void*)d.data, (void*)s.data, (void*)r.data, (void*)o.data);
        fill(a, 1, -1.0, 1.0);
        fill(d, 2, 4.0, 5.0);
        tridiag_stack_variant<32, 40, 8, false>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_variant<double, 1, 8>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    }
    lap_suite(512, 512, 512, 0, "512^3");
}

// Which (I tile, J row) a workgroup takes (tridiag_pipe_kernel's MAP): time, and -- under rocprofv3 --pmc -- the translation
// counters, per mapping (round 4).  Also on a domain that is no multiple of anything.
static void section_trimap() {
    const int reps = getenv("MB_TRIMAP_REPS") ? atoi(getenv("MB_TRIMAP_REPS")) : 2;
    for (int rep = 0; rep < reps; ++rep) {
        const int dI = 1024, dJ = 1024, dK = 160;
        void* dummy = nullptr;  // MB_TRIMAP_SHUFFLE: another placement of the fields in every pass
        if (getenv("MB_TRIMAP_SHUFFLE")) CK(hipMalloc(&dummy, (size_t)(rep * 37 + 5) * (2u << 20) * (size_t)atoi(getenv("MB_TRIMAP_SHUFFLE"))));
        const int xp = getenv("MB_TRI_EXTRA_PITCH") ? atoi(getenv("MB_TRI_EXTRA_PITCH")) : 0;  // items added to the row pitch: the K stride is no power of two
        DevField<double> a(dI, dJ, dK, 0, 0, 32, xp), d(dI, dJ, dK, 0, 0, 32, xp), s(dI, dJ, dK, 0, 0, 32, xp), r(dI, dJ, dK, 0, 0, 32, xp), o(dI, dJ, dK, 0, 0, 32, xp);
        DevField<double> s2(dI, dJ, dK, 0, 0, 32, xp), r2(dI, dJ, dK, 0, 0, 32, xp), o2(dI, dJ, dK, 0, 0, 32, xp);
        if (dummy) hipFree(dummy);
        fill(a, 1, -1.0, 1.0);
        fill(d, 2, 4.0, 5.0);
        printf("trimap     fields (MiB, modulo 1 GiB): inf %.3f diag %.3f sup %.3f rhs %.3f out %.3f   raw inf %p\n",
               (reinterpret_cast<uintptr_t>(a.data) % (1ull << 30)) / 1048576.0, (reinterpret_cast<uintptr_t>(d.data) % (1ull << 30)) / 1048576.0,
               (reinterpret_cast<uintptr_t>(s.data) % (1ull << 30)) / 1048576.0, (reinterpret_cast<uintptr_t>(r.data) % (1ull << 30)) / 1048576.0,
               (reinterpret_cast<uintptr_t>(o.data) % (1ull << 30)) / 1048576.0, (void*)a.raw);
        tridiag_stack_variant<104, 40, 4, true, 1, 0>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        if (getenv("MB_TRIMAP_ONLY0")) continue;
        tridiag_stack_variant<104, 40, 4, true, 1, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 1, 2>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 1, 3>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<80, 40, 8, true, 1, 0>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<80, 40, 8, true, 1, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<80, 40, 8, true, 1, 2>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<80, 40, 8, true, 1, 3>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    }
    {
        const int dI = 200, dJ = 301, dK = 150;
        DevField<double> a(dI, dJ, dK, 0, 0), d(dI, dJ, dK, 0, 0), s(dI, dJ, dK, 0, 0), r(dI, dJ, dK, 0, 0), o(dI, dJ, dK, 0, 0);
        DevField<double> s2(dI, dJ, dK, 0, 0), r2(dI, dJ, dK, 0, 0), o2(dI, dJ, dK, 0, 0);
        fill(a, 1, -1.0, 1.0);
        fill(d, 2, 4.0, 5.0);
        tridiag_stack_variant<104, 40, 4, true, 1, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 1, 2>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 1, 3>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    }
}

// Round 5: is the speed of the K-strided column kernels a property of the individual ALLOCATION (profiles/r4_tridiag_translation.txt:
// the tridiagonal solve runs at 0.60-0.73 of the peak depending on which allocations its five fields live in)?  A probe reads ONE
// field the way the solve's forward sweep does -- a wave per 64 columns of a row, all K levels one plane apart, 512 bytes per level --
// and is timed on each of N separately hipMalloc'ed fields; then the solve runs on the five fields the probe found fastest and on the
// five it found slowest.  If the probe separates the allocations and the solve follows, an allocator can probe and choose.
__global__ void __launch_bounds__(64) kcol_probe_kernel(const double* __restrict__ p, int64_t sj, int64_t sk, int dK, unsigned ti, double* __restrict__ sink) {
    const unsigned tile = blockIdx.x % ti, j = blockIdx.x / ti;
    const double* q = p + (int64_t)j * sj + (int64_t)tile * 64 + threadIdx.x;
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < dK; ++k) acc += q[(int64_t)k * sk];
    if (acc == 12345.678) sink[0] = acc;  // (never: the loads must not be optimised away)
}

// ... and the same columns written (MODE 1) or read, scaled and written back in place (MODE 2: what the forward sweep does to sup and rhs)
template <int MODE>
__global__ void __launch_bounds__(64) kcol_write_probe_kernel(double* __restrict__ p, int64_t sj, int64_t sk, int dK, unsigned ti) {
    const unsigned tile = blockIdx.x % ti, j = blockIdx.x / ti;
    double* q = p + (int64_t)j * sj + (int64_t)tile * 64 + threadIdx.x;
#pragma unroll 8
    for (int k = 0; k < dK; ++k) {
        if (MODE == 1) q[(int64_t)k * sk] = 4.5;
        else q[(int64_t)k * sk] = q[(int64_t)k * sk] * 1.0000001;
    }
}

__global__ void __launch_bounds__(256) stream_write_probe_kernel(u32x4* __restrict__ p, long n) {
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) __builtin_nontemporal_store(v, p + t);
}

// MB_KPROBE_MAP=N: N separately allocated fields of 1.34 GB, each probed with the K-strided write, a K-strided read and a plain
// streaming write -- a map of how the device's memory answers, allocation by allocation (no solve).
static void kprobe_map(int nb) {
    const int dI = 1024, dJ = 1024, dK = 160;
    const unsigned ti = (unsigned)cdiv(dI, 64);
    std::vector<DevField<double>*> f;
    double* sink = nullptr;
    CK(hipMalloc(&sink, 64));
    for (int b = 0; b < nb; ++b) {
        f.push_back(new DevField<double>(dI, dJ, dK, 0, 0));
        DevField<double>& x = *f.back();
        fill(x, 100 + b, 4.0, 5.0);
        const double gb = (double)dI * dJ * dK * 8.0 / 1e9;
        const double w = time_ms([&](int) { hipLaunchKernelGGL(kcol_write_probe_kernel<1>, dim3(ti * dJ), dim3(64), 0, 0, x.data, x.sj, x.sk, dK, ti); }, 6, 2);
        const double r = time_ms([&](int) { hipLaunchKernelGGL(kcol_probe_kernel, dim3(ti * dJ), dim3(64), 0, 0, x.data, x.sj, x.sk, dK, ti, sink); }, 6, 2);
        const double sw = time_ms([&](int) { hipLaunchKernelGGL(stream_write_probe_kernel, dim3(8192), dim3(256), 0, 0, (u32x4*)x.data, (long)((size_t)dI * dJ * dK / 2)); }, 6, 2);
        printf("kmap       field %3d at %p (+%7.1f MiB from the previous): K-strided write %7.1f GB/s  K-strided read %7.1f GB/s  streaming write %7.1f GB/s  %s\n", b,
               (void*)x.raw, b ? ((double)((intptr_t)f[b - 1]->raw - (intptr_t)x.raw)) / 1048576.0 : 0.0, gb / (w * 1e-3), gb / (r * 1e-3), gb / (sw * 1e-3),
               gb / (w * 1e-3) > 5800.0 ? "FAST" : "slow");
        fflush(stdout);
    }
    for (auto* x : f) delete x;
    CK(hipFree(sink));
}

// MB_KPAIRS=N: do two fields that are written side by side (sup and rhs in the solve's forward sweep) get in each other's way, and does
// that depend on WHICH two allocations they are?  N fields; every pair (a, b) is written K-strided in ONE kernel (two stores per level),
// and every field alone; printed as a matrix of GB/s.  A pairwise effect (DRAM banks shared by two streams whose addresses collide in
// the bank bits) would show as pairs that are slow although both members are fast alone.
__global__ void __launch_bounds__(64) kcol_pair_write_probe_kernel(double* __restrict__ p, double* __restrict__ q2, int64_t sj, int64_t sk, int dK, unsigned ti) {
    const unsigned tile = blockIdx.x % ti, j = blockIdx.x / ti;
    const int64_t off = (int64_t)j * sj + (int64_t)tile * 64 + threadIdx.x;
    double *a = p + off, *b = q2 + off;
#pragma unroll 8
    for (int k = 0; k < dK; ++k) {
        a[(int64_t)k * sk] = 4.5;
        b[(int64_t)k * sk] = 5.5;
    }
}

static void kpairs(int n) {
    const int dI = 1024, dJ = 1024, dK = 160;
    const unsigned ti = 16;
    std::vector<DevField<double>*> f;
    for (int b = 0; b < n; ++b) {
        f.push_back(new DevField<double>(dI, dJ, dK, 0, 0));
        fill(*f.back(), 100 + b, 4.0, 5.0);
    }
    const double gb = (double)dI * dJ * dK * 8.0 / 1e9;
    printf("kpairs     K-strided writes, GB/s: diagonal = the field alone, (a, b) = both fields written in one kernel (bytes of both counted)\n");
    for (int a = 0; a < n; ++a) {
        printf("kpairs     %2d |", a);
        for (int b = 0; b < n; ++b) {
            double ms;
            if (a == b) {
                ms = time_ms([&](int) { hipLaunchKernelGGL(kcol_write_probe_kernel<1>, dim3(ti * dJ), dim3(64), 0, 0, f[a]->data, f[a]->sj, f[a]->sk, dK, ti); }, 8, 2);
                printf(" [%5.0f]", gb / (ms * 1e-3));
            } else if (b > a) {
                ms = time_ms([&](int) { hipLaunchKernelGGL(kcol_pair_write_probe_kernel, dim3(ti * dJ), dim3(64), 0, 0, f[a]->data, f[b]->data, f[a]->sj, f[a]->sk, dK, ti); }, 8, 2);
                printf("  %5.0f ", 2.0 * gb / (ms * 1e-3));
            } else {
                printf("    .   ");
            }
        }
        printf("\n");
    }
    fflush(stdout);
    // ---- groups: a and b are in the same group when writing both is no faster than 6.55 TB/s --------------------------------------
    std::vector<std::vector<double>> pair(n, std::vector<double>(n, 0.0));
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) {
            const double ms = time_ms([&](int) { hipLaunchKernelGGL(kcol_pair_write_probe_kernel, dim3(ti * dJ), dim3(64), 0, 0, f[a]->data, f[b]->data, f[a]->sj, f[a]->sk, dK, ti); }, 8, 2);
            pair[a][b] = pair[b][a] = 2.0 * gb / (ms * 1e-3);
        }
    std::vector<int> group(n, -1);
    int ngroups = 0;
    for (int a = 0; a < n; ++a) {
        if (group[a] >= 0) continue;
        group[a] = ngroups++;
        for (int b = a + 1; b < n; ++b)
            if (group[b] < 0 && pair[a][b] < 6550.0) group[b] = group[a];
    }
    printf("kpairs     groups:");
    for (int a = 0; a < n; ++a) printf(" %d:%c", a, 'A' + group[a]);
    printf("\n");
    auto pick = [&](int g, int nth) {  // the nth field of group g, or -1
        for (int a = 0; a < n; ++a)
            if (group[a] == g && nth-- == 0) return a;
        return -1;
    };
    // the headline stencil: in and out in ONE group against in and out in DIFFERENT groups
    {
        const int64_t d[3] = {512, 512, 512};
        const int64_t sj = 528, sk = sj * 514;
        auto lap_ms = [&](int fi, int fo) {
            const View<const double> in{f[fi]->data + 16 + sj, 1, sj, sk};
            const View<double> out{f[fo]->data + 16 + sj, 1, sj, sk};
            return time_ms([&](int) { (void)lap5_launch_variant<double, double, GT4MI_LAP_NOTEBOOK>(in, out, d, 0); }, 60, 5);
        };
        const double lups = 512.0 * 512.0 * 512.0;
        for (int g = 0; g < ngroups; ++g) {
            const int a = pick(g, 0), b = pick(g, 1);
            if (a >= 0 && b >= 0) {
                const double ms = lap_ms(a, b);
                printf("kpairs     Laplacian 512^3  in %2d  out %2d  (both group %c)        %.4f ms  %.1f GLUPS  %.3f of 8 TB/s\n", a, b, 'A' + g, ms, lups / ms / 1e6, 16.0 * lups / (ms * 1e-3) / 8e12);
            }
            for (int h = 0; h < ngroups; ++h) {
                const int c = pick(h, 0);
                if (h == g || a < 0 || c < 0) continue;
                const double ms = lap_ms(a, c);
                printf("kpairs     Laplacian 512^3  in %2d  out %2d  (groups %c -> %c)       %.4f ms  %.1f GLUPS  %.3f of 8 TB/s\n", a, c, 'A' + g, 'A' + h, ms, lups / ms / 1e6, 16.0 * lups / (ms * 1e-3) / 8e12);
            }
        }
    }
    // horizontal diffusion, the configs[4] share (512 x 1024 x 80 fp64, 339 MB per field) and configs[2] (fp32): in / coeff / out in one
    // group against in + out in one and coeff in another, and against out alone in the other
    if (ngroups >= 2) {
        const int64_t d[3] = {512, 1024, 80};
        const int64_t sj = 528, sk = sj * 1028;
        auto views = [&](int fi, int fc, int fo, View<const double>* in, View<const double>* cf, View<double>* out) {
            *in = View<const double>{f[fi]->data + 16 + 2 * sj, 1, sj, sk};
            *cf = View<const double>{f[fc]->data + 16 + 2 * sj, 1, sj, sk};
            *out = View<double>{f[fo]->data + 16 + 2 * sj, 1, sj, sk};
        };
        auto hd_ms = [&](int fi, int fc, int fo) {
            View<const double> in, cf;
            View<double> out;
            views(fi, fc, fo, &in, &cf, &out);
            return time_ms([&](int) { (void)hdiff_launch<double, double, double, true, true>(in, out, cf, 0.0, d, 0); }, 100, 10);
        };
        const int a0 = pick(0, 0), a1 = pick(0, 1), a2 = pick(0, 2), b0 = pick(1, 0);
        if (a0 >= 0 && a1 >= 0 && a2 >= 0 && b0 >= 0) {
            for (int rep = 0; rep < 2; ++rep) {
                const double same = hd_ms(a0, a1, a2), cf_other = hd_ms(a0, b0, a2), out_other = hd_ms(a0, a1, b0), in_other = hd_ms(b0, a1, a2);
                const double bytes = 24.0 * 512 * 1024 * 80;
                printf("kpairs     hdiff fp64 512x1024x80  all in one group %.4f ms (%.3f)   coeff in the other %.4f (%.3f)   out in the other %.4f (%.3f)   in in the other %.4f (%.3f of 8 TB/s)\n",
                       same, bytes / (same * 1e-3) / 8e12, cf_other, bytes / (cf_other * 1e-3) / 8e12, out_other, bytes / (out_other * 1e-3) / 8e12, in_other, bytes / (in_other * 1e-3) / 8e12);
            }
        }
    }
    // the solve: inf, diag, sup, rhs, out dealt over the groups round-robin against all five from the largest group
    if (n >= 10) {
        int largest = 0;
        std::vector<int> count(ngroups, 0);
        for (int a = 0; a < n; ++a) ++count[group[a]];
        for (int g = 1; g < ngroups; ++g)
            if (count[g] > count[largest]) largest = g;
        std::vector<int> spread, same, used(n, 0);
        for (int i = 0, nth = 0; (int)spread.size() < 8 && nth < n; ++i) {
            const int a = pick(i % ngroups, i / ngroups);
            if (i % ngroups == ngroups - 1) ++nth;
            if (a >= 0 && !used[a]) { spread.push_back(a); used[a] = 1; }
        }
        for (int a = 0; a < n && (int)same.size() < 8; ++a)
            if (group[a] == largest) same.push_back(a);
        auto solve = [&](const char* what, const std::vector<int>& v) {
            if (v.size() < 8) { printf("kpairs     (not enough fields for the solve %s)\n", what); return; }
            printf("kpairs     the solve with inf diag sup rhs out = fields %d %d %d %d %d (%s):\n", v[0], v[1], v[2], v[3], v[4], what);
            fill(*f[v[0]], 1, -1.0, 1.0);
            fill(*f[v[1]], 2, 4.0, 5.0);
            tridiag_stack_variant<104, 40, 4, true, 1, 0>(*f[v[0]], *f[v[1]], *f[v[2]], *f[v[3]], *f[v[4]], *f[v[5]], *f[v[6]], *f[v[7]], dI, dJ, dK);
        };
        solve("dealt over the groups round-robin", spread);
        if (count[largest] >= 8) solve("all in one group", same);
        solve("dealt over the groups round-robin", spread);
    }
    fflush(stdout);
    for (auto* x : f) delete x;
}

// MB_KCHUNK=N: is the write speed a property of the PHYSICAL memory or of how it is mapped?  N physical chunks of 256 MiB
// (hipMemCreate), each mapped at a 256-MiB-aligned address, probed (K-strided write over its 32 planes of 8 MiB, and a streaming write),
// unmapped, mapped again 2 MiB further (an address that is NOT aligned to the chunk), probed again, and a third time at the first address.
static void kchunk(int n) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = (size_t)256 << 20, align = chunk;
    void* reserved = nullptr;
    CK(hipMemAddressReserve(&reserved, 4 * chunk, align, nullptr, 0));
    char* base = (char*)(((uintptr_t)reserved + align - 1) / align * align);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const int dI = 1024, dJ = 1024, dK = 32;
    const unsigned ti = 16;
    auto probe = [&](char* va, double* kw, double* sw) {
        double* p = (double*)va;
        const double gb = (double)chunk / 1e9;
        const double a = time_ms([&](int) { hipLaunchKernelGGL(kcol_write_probe_kernel<1>, dim3(ti * dJ), dim3(64), 0, 0, p, (int64_t)dI, (int64_t)dI * dJ, dK, ti); }, 20, 3);
        const double b = time_ms([&](int) { hipLaunchKernelGGL(stream_write_probe_kernel, dim3(8192), dim3(256), 0, 0, (u32x4*)p, (long)(chunk / 16)); }, 20, 3);
        *kw = gb / (a * 1e-3);
        *sw = gb / (b * 1e-3);
    };
    std::vector<hipMemGenericAllocationHandle_t> held;
    for (int c = 0; c < n; ++c) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        held.push_back(h);  // (kept: the next chunk is another piece of physical memory)
        double kw[3], sw[3];
        char* where[3] = {base, base + chunk + ((size_t)2 << 20), base};
        for (int t = 0; t < 3; ++t) {
            CK(hipMemMap(where[t], chunk, 0, h, 0));
            CK(hipMemSetAccess(where[t], chunk, &acc, 1));
            probe(where[t], &kw[t], &sw[t]);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(where[t], chunk));
        }
        printf("kchunk     chunk %3d: aligned VA  K-strided write %7.1f  streaming write %7.1f | VA + 2 MiB  %7.1f  %7.1f | aligned again  %7.1f  %7.1f GB/s  %s\n", c, kw[0], sw[0],
               kw[1], sw[1], kw[2], sw[2], kw[0] > 5800.0 ? "FAST" : "slow");
        fflush(stdout);
    }
    for (auto h : held) (void)hipMemRelease(h);
    (void)hipMemAddressFree(reserved, 4 * chunk);
}

static void section_kprobe() {
    if (getenv("MB_KCHUNK")) {
        kchunk(atoi(getenv("MB_KCHUNK")));
        return;
    }
    if (getenv("MB_KPAIRS")) {
        kpairs(atoi(getenv("MB_KPAIRS")));
        return;
    }
    if (getenv("MB_KPROBE_MAP")) {
        kprobe_map(atoi(getenv("MB_KPROBE_MAP")));
        return;
    }
    const int dI = 1024, dJ = 1024, dK = 160;
    const int NB = getenv("MB_KPROBE_FIELDS") ? atoi(getenv("MB_KPROBE_FIELDS")) : 14;
    std::vector<DevField<double>*> f;
    for (int b = 0; b < NB; ++b) {
        f.push_back(new DevField<double>(dI, dJ, dK, 0, 0));
        fill(*f.back(), 100 + b, 4.0, 5.0);
    }
    double* sink = nullptr;
    CK(hipMalloc(&sink, 64));
    const unsigned ti = (unsigned)cdiv(dI, 64);
    std::vector<std::pair<double, int>> order;
    for (int pass = 0; pass < 2; ++pass)
        for (int b = 0; b < NB; ++b) {
            const double ms = time_ms([&](int) {
                hipLaunchKernelGGL(kcol_probe_kernel, dim3(ti * dJ), dim3(64), 0, 0, f[b]->data, f[b]->sj, f[b]->sk, dK, ti, sink);
            }, 10, 2);
            printf("kprobe     pass %d field %2d at %p (MiB mod 1 GiB %9.3f): %8.4f ms  %7.1f GB/s\n", pass, b, (void*)f[b]->raw,
                   (reinterpret_cast<uintptr_t>(f[b]->data) % (1ull << 30)) / 1048576.0, ms, (double)dI * dJ * dK * 8.0 / (ms * 1e-3) / 1e9);
            if (pass == 1) order.push_back({ms, b});
        }
    for (int b = 0; b < NB; ++b) {
        const double w = time_ms([&](int) { hipLaunchKernelGGL(kcol_write_probe_kernel<1>, dim3(ti * dJ), dim3(64), 0, 0, f[b]->data, f[b]->sj, f[b]->sk, dK, ti); }, 10, 2);
        const double rw = time_ms([&](int) { hipLaunchKernelGGL(kcol_write_probe_kernel<2>, dim3(ti * dJ), dim3(64), 0, 0, f[b]->data, f[b]->sj, f[b]->sk, dK, ti); }, 10, 2);
        const double gb = (double)dI * dJ * dK * 8.0 / 1e9;
        printf("kprobe     field %2d: K-strided write %8.4f ms %7.1f GB/s   read-modify-write in place %8.4f ms %7.1f GB/s (read + written)\n", b, w, gb / (w * 1e-3),
               rw, 2.0 * gb / (rw * 1e-3));
        if (getenv("MB_KPROBE_BY_WRITE")) order[b] = {w, b};  // (rank the fields by the WRITE probe instead)
    }
    std::sort(order.begin(), order.end());
    printf("kprobe     fastest -> slowest:");
    for (auto& o : order) printf(" %d(%.3f)", o.second, o.first);
    printf("\n");
    fflush(stdout);
    auto solve_on = [&](const char* what, int i0, int step) {
        DevField<double>*a = f[order[i0].second], *d = f[order[i0 + step].second], *s = f[order[i0 + 2 * step].second],
                        *r = f[order[i0 + 3 * step].second], *o = f[order[i0 + 4 * step].second];
        // (the check copies: the fields in the middle of the order)
        const int mid = NB / 2;
        DevField<double>*s2 = f[order[mid - 1].second], *r2 = f[order[mid].second], *o2 = f[order[mid + 1].second];
        fill(*a, 1, -1.0, 1.0);
        fill(*d, 2, 4.0, 5.0);
        printf("kprobe     the solve on the five %s fields (%d %d %d %d %d):\n", what, order[i0].second, order[i0 + step].second,
               order[i0 + 2 * step].second, order[i0 + 3 * step].second, order[i0 + 4 * step].second);
        for (int rep = 0; rep < 2; ++rep) tridiag_stack_variant<104, 40, 4, true, 1, 0>(*a, *d, *s, *r, *o, *s2, *r2, *o2, dI, dJ, dK);
    };
    if (NB >= 13 && !getenv("MB_KPROBE_NO_SOLVE")) {
        solve_on("FASTEST", 0, 1);
        solve_on("SLOWEST", NB - 1, -1);
        solve_on("FASTEST", 0, 1);
    }
    if (NB >= 4 && getenv("MB_KPROBE_LAP")) {
        // the headline stencil with its fields inside allocations of either class: 512^3 fp64, rows of 528 items, origin (1, 1)
        const int64_t d[3] = {512, 512, 512};
        const int64_t sj = 528, sk = sj * 514;
        auto lap_ms = [&](DevField<double>* fi, DevField<double>* fo) {
            const View<const double> in{fi->data + 16 + sj, 1, sj, sk};
            const View<double> out{fo->data + 16 + sj, 1, sj, sk};
            return time_ms([&](int) { (void)lap5_launch_variant<double, double, GT4MI_LAP_NOTEBOOK>(in, out, d, 0); }, 60, 5);
        };
        DevField<double>*fast_a = f[order[0].second], *fast_b = f[order[1].second], *slow_a = f[order[NB - 1].second], *slow_b = f[order[NB - 2].second];
        for (int rep = 0; rep < 2; ++rep) {
            const double ss = lap_ms(slow_a, slow_b), sf = lap_ms(slow_a, fast_a), ff = lap_ms(fast_b, fast_a), fs = lap_ms(fast_b, slow_b);
            const double lups = 512.0 * 512.0 * 512.0;
            printf("kprobe     Laplacian 512^3, in / out in allocations of class  slow/slow %.4f ms (%.1f GLUPS)  slow/FAST %.4f (%.1f)  FAST/FAST %.4f (%.1f)  FAST/slow %.4f (%.1f)\n",
                   ss, lups / ss / 1e6, sf, lups / sf / 1e6, ff, lups / ff / 1e6, fs, lups / fs / 1e6);
        }
    }
    for (auto* x : f) delete x;
    CK(hipFree(sink));
}

// pipelined vs plain on-chip-stack kernel, also on column counts / depths that exercise the head and odd-batch paths
// trint: nontemporal LOADS in the solve (round 5; the Laplacian lost 6-9 % with them in round 1 -- its halo rows are re-read through
// L1 --, but a column kernel reads every element exactly once).  A-B-A in one process, same fields.
static void section_trint() {
    const int dI = 1024, dJ = 1024, dK = 160;
    DevField<double> a(dI, dJ, dK, 0, 0), d(dI, dJ, dK, 0, 0), s(dI, dJ, dK, 0, 0), r(dI, dJ, dK, 0, 0), o(dI, dJ, dK, 0, 0);
    DevField<double> s2(dI, dJ, dK, 0, 0), r2(dI, dJ, dK, 0, 0), o2(dI, dJ, dK, 0, 0);
    fill(a, 1, -1.0, 1.0);
    fill(d, 2, 4.0, 5.0);
    for (int rep = 0; rep < 3; ++rep) {
        tridiag_stack_variant<104, 40, 4, true, 1, 0, 0>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 1, 0, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 1, 0, 2>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        if (getenv("MB_TRINT_SHAPES")) {  // do the batch / depth optima move with nontemporal loads?
            tridiag_stack_variant<96, 40, 4, true, 1, 0, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 2, true, 1, 0, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 2, 0, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 1, 1, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 2, 1, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 4, 0, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 4, 1, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 2, 2, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 2, 3, 1>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
            tridiag_stack_variant<104, 40, 4, true, 2, 0, 0>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        }
    }
    tridiag_stack_variant<104, 40, 4, true, 1, 0, 0>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    if (getenv("MB_TRINT_SHALLOW")) {  // the variants of shallower columns (K = 80: 32 + 40 levels, K = 60: 16 + 40): one or two waves per workgroup
        for (int K : {80, 60}) {
            DevField<double> a2(dI, dJ, K, 0, 0), d2(dI, dJ, K, 0, 0), s3(dI, dJ, K, 0, 0), r3(dI, dJ, K, 0, 0), o3(dI, dJ, K, 0, 0);
            DevField<double> s4(dI, dJ, K, 0, 0), r4(dI, dJ, K, 0, 0), o4(dI, dJ, K, 0, 0);
            fill(a2, 1, -1.0, 1.0);
            fill(d2, 2, 4.0, 5.0);
            for (int rep = 0; rep < 3; ++rep) {
                if (K == 80) {
                    tridiag_stack_variant<32, 40, 8, true, 1, 0, 1>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                    tridiag_stack_variant<32, 40, 8, true, 2, 0, 1>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                    tridiag_stack_variant<32, 40, 8, true, 1, 0, 0>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                    tridiag_stack_variant<32, 40, 4, true, 1, 0, 1>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                } else {
                    tridiag_stack_variant<16, 40, 8, true, 1, 0, 1>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                    tridiag_stack_variant<16, 40, 8, true, 2, 0, 1>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                    tridiag_stack_variant<16, 40, 8, true, 1, 0, 0>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                    tridiag_stack_variant<16, 40, 4, true, 1, 0, 1>(a2, d2, s3, r3, o3, s4, r4, o4, dI, dJ, K);
                }
            }
        }
    }
}

static void section_tripipe() {
    for (int rep = 0; rep < 2; ++rep) {
        const int dI = 1024, dJ = 1024, dK = 160;
        DevField<double> a(dI, dJ, dK, 0, 0), d(dI, dJ, dK, 0, 0), s(dI, dJ, dK, 0, 0), r(dI, dJ, dK, 0, 0), o(dI, dJ, dK, 0, 0);
        DevField<double> s2(dI, dJ, dK, 0, 0), r2(dI, dJ, dK, 0, 0), o2(dI, dJ, dK, 0, 0);
        fill(a, 1, -1.0, 1.0);
        fill(d, 2, 4.0, 5.0);
        tridiag_stack_variant<32, 40, 8, false>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 40, 4, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<16, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<24, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<40, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<48, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<64, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<80, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<56, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 32, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 16, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<16, 16, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 0, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 48, 16, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<80, 40, 4, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<96, 40, 4, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<112, 40, 4, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 2>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<104, 40, 4, true, 4>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<80, 40, 8, true, 4>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 40, 8, true, 4>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    }
    for (int dK : {73, 74, 80, 81, 87, 88, 89, 96, 97, 105, 33, 34, 40, 41, 47, 48, 49, 72}) {  // every head / parity case
        const int dI = 200, dJ = 37;
        DevField<double> a(dI, dJ, dK, 0, 0), d(dI, dJ, dK, 0, 0), s(dI, dJ, dK, 0, 0), r(dI, dJ, dK, 0, 0), o(dI, dJ, dK, 0, 0);
        DevField<double> s2(dI, dJ, dK, 0, 0), r2(dI, dJ, dK, 0, 0), o2(dI, dJ, dK, 0, 0);
        fill(a, 1, -1.0, 1.0);
        fill(d, 2, 4.0, 5.0);
        printf("           dK = %d\n", dK);
        tridiag_stack_variant<32, 40, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 0, 8, true>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
        tridiag_stack_variant<32, 40, 8, true, 4>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    }
}


// ---- vertical advection with the top of the column on chip (vadv_stack.hip.h) ---------------------
struct VadvSet {
    DevField<double> wcon, u_stage, u_pos, utens, ts0, ts, ts_ref, ccol, dcol;
    int dI, dJ, dK;
    VadvSet(int dI_, int dJ_, int dK_)
        : wcon(dI_, dJ_, dK_, 1, 0), u_stage(dI_, dJ_, dK_, 0, 0), u_pos(dI_, dJ_, dK_, 0, 0), utens(dI_, dJ_, dK_, 0, 0),
          ts0(dI_, dJ_, dK_, 0, 0), ts(dI_, dJ_, dK_, 0, 0), ts_ref(dI_, dJ_, dK_, 0, 0), ccol(dI_, dJ_, dK_, 0, 0),
          dcol(dI_, dJ_, dK_, 0, 0), dI(dI_), dJ(dJ_), dK(dK_) {
        fill(wcon, 11, -1.0, 1.0);
        fill(u_stage, 12, -1.0, 1.0);
        fill(u_pos, 13, -1.0, 1.0);
        fill(utens, 14, -1.0, 1.0);
        fill(ts0, 15, -1.0, 1.0);
    }
    VadvFields fields(DevField<double>& out) {
        return VadvFields{wcon.cview(), u_stage.cview(), u_pos.cview(), utens.cview(), out.view(), ccol.view(), dcol.view()};
    }
    void reset(DevField<double>& out) { CK(hipMemcpy(out.raw, ts0.raw, ts0.bytes, hipMemcpyDeviceToDevice)); }
};

static constexpr double VADV_DTR = 3.0 / 20.0;

static void vadv_plain(VadvSet& s, DevField<double>& out) {
    hipLaunchKernelGGL(vadv_plain_kernel, dim3((unsigned)cdiv(s.dI, 64), (unsigned)cdiv(s.dJ, 4)), dim3(64, 4), 0, 0, s.fields(out),
                       VADV_DTR, 0.5, 0.5, s.dI, s.dJ, s.dK);
}

template <int RL, int LL, int U, bool PIPE = true, bool SADDR = true, int SLOTS = 0, int NTM = 0>
static void vadv_variant(VadvSet& s, bool time_it = true) {
    const unsigned tiles_i = (unsigned)cdiv(s.dI, 64);
    auto launch = [&](DevField<double>& out) {
        hipLaunchKernelGGL((vadv_pipe_kernel<RL, LL, U, PIPE, SADDR, SLOTS, NTM>), dim3(tiles_i * (unsigned)s.dJ), dim3(64), 0, 0, s.fields(out), VADV_DTR, 0.5,
                           0.5, s.dI, s.dJ, s.dK, tiles_i);
    };
    char cfg[96];
    snprintf(cfg, sizeof cfg, "%dx%dx%d regs %d lds %d batch %d%s%s slots %d nt %d", s.dI, s.dJ, s.dK, RL, LL, U, PIPE ? "" : " nopipe",
             SADDR ? "" : " vaddr", SLOTS, NTM);
    if (s.dK - RL - LL < 1) {
        printf("vadv       %-44s skipped (needs dK > %d)\n", cfg, RL + LL);
        return;
    }
    s.reset(s.ts);
    launch(s.ts);
    CK(hipDeviceSynchronize());
    const unsigned long long bad = count_diff(s.ts, s.ts_ref, s.dI, s.dJ, s.dK);
    if (bad) printf("vadv       %-44s MISMATCH in %llu points\n", cfg, bad);
    if (time_it) {
        const double ms = time_ms([&](int) { launch(s.ts); }, 20);
        report(bad ? "vadv BAD" : "vadv", cfg, ms, (double)s.dI * s.dJ * s.dK, 48.0);
    } else if (!bad) {
        printf("vadv       %-44s identical\n", cfg);
    }
}

static void section_vadv() {
    if (getenv("MB_VADV_NT")) {  // nontemporal loads in the hand-written restatement: mode 5's placement, and wcon through a lane shift
        VadvSet s(1024, 1024, 160);
        s.reset(s.ts_ref);
        vadv_plain(s, s.ts_ref);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 4; ++rep) {
            vadv_variant<104, 40, 4, true, true, 0, 0>(s);
            vadv_variant<104, 40, 4, true, true, 0, 1>(s);
            vadv_variant<104, 40, 4, true, true, 0, 2>(s);
        }
        return;
    }
    if (getenv("MB_VADV_SLOTS")) {  // where do the 16 spilled levels go: the column's own place in a 1.3 GB array, or a small reused area?
        VadvSet s(1024, 1024, 160);
        s.reset(s.ts_ref);
        vadv_plain(s, s.ts_ref);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 3; ++rep) {
            vadv_variant<104, 40, 4, true, true, 0>(s);
            vadv_variant<104, 40, 4, true, true, 2048>(s);
            vadv_variant<104, 40, 4, true, true, 4096>(s);
            vadv_variant<104, 40, 4, true, true, 8192>(s);
            vadv_variant<80, 40, 4, true, true, 0>(s);
            vadv_variant<80, 40, 4, true, true, 4096>(s);
            vadv_variant<48, 40, 4, true, true, 0>(s);
            vadv_variant<48, 40, 4, true, true, 4096>(s);
        }
        return;
    }
    {
        VadvSet s(1024, 1024, 160);
        s.reset(s.ts_ref);
        vadv_plain(s, s.ts_ref);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep) {
            const double ms = time_ms([&](int) { vadv_plain(s, s.ts); }, 20);
            report("vadv", "1024x1024x160 plain (no cache, 64x4 blocks)", ms, (double)s.dI * s.dJ * s.dK, 48.0);
            vadv_variant<16, 40, 4>(s);
            vadv_variant<16, 40, 8>(s);
            vadv_variant<32, 40, 4>(s);
            vadv_variant<32, 40, 8>(s);
            vadv_variant<48, 40, 4>(s);
            vadv_variant<48, 40, 8>(s);
            vadv_variant<64, 40, 4>(s);
            vadv_variant<64, 40, 8>(s);
            vadv_variant<80, 40, 4>(s);
            vadv_variant<80, 40, 8>(s);
            vadv_variant<96, 40, 4>(s);
            vadv_variant<104, 40, 4>(s);
            vadv_variant<112, 40, 4>(s);
            vadv_variant<104, 40, 4, false, true>(s);
            vadv_variant<104, 40, 4, true, false>(s);
            vadv_variant<104, 40, 4, false, false>(s);
            vadv_variant<16, 40, 4, false, true>(s);
            vadv_variant<16, 40, 4, true, false>(s);
            vadv_variant<64, 0, 4>(s);
            vadv_variant<32, 0, 8>(s);
        }
    }
    for (int dK : {58, 59, 60, 61, 62, 63, 64, 65, 66, 73, 80, 81}) {  // every head / parity / leftover case
        VadvSet s(200, 37, dK);
        s.reset(s.ts_ref);
        vadv_plain(s, s.ts_ref);
        CK(hipDeviceSynchronize());
        vadv_variant<16, 40, 4>(s, false);
        vadv_variant<16, 40, 8>(s, false);
        vadv_variant<16, 0, 8>(s, false);
        vadv_variant<8, 0, 4>(s, false);
    }
}

static void section_tridiag() {
    const int dI = 1024, dJ = 1024, dK = 160;
    DevField<double> a(dI, dJ, dK, 0, 0), d(dI, dJ, dK, 0, 0), s(dI, dJ, dK, 0, 0), r(dI, dJ, dK, 0, 0), o(dI, dJ, dK, 0, 0);
    DevField<double> s2(dI, dJ, dK, 0, 0), r2(dI, dJ, dK, 0, 0), o2(dI, dJ, dK, 0, 0);
    fill(a, 1, -1.0, 1.0);
    fill(d, 2, 4.0, 5.0);
    // correctness: vector kernel vs any-stride kernel from identical inputs
    fill(s, 3, -1.0, 1.0);
    fill(r, 4, -10.0, 10.0);
    fill(s2, 3, -1.0, 1.0);
    fill(r2, 4, -10.0, 10.0);
    {
        const unsigned ti = (unsigned)cdiv(dI, 512);
        hipLaunchKernelGGL((tridiag_kernel<double, 2, 4>), dim3(ti * dJ), dim3(256), 0, 0, a.cview(), d.cview(), s.view(), r.view(), o.view(), dI, dJ, dK, ti);
        dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4));
        const double ms = time_ms([&](int) {
            fill(s2, 3, -1.0, 1.0);
            fill(r2, 4, -10.0, 10.0);
            hipLaunchKernelGGL((tridiag_generic_kernel<double>), grid, dim3(256), 0, 0, a.cview(), d.cview(), s2.view(), r2.view(), o2.view(), dI, dJ, dK);
        }, 1, 0);
        (void)ms;
        CK(hipDeviceSynchronize());
        printf("           check vec2 vs generic: out %llu sup %llu rhs %llu mismatches\n", count_diff(o, o2, dI, dJ, dK),
               count_diff(s, s2, dI, dJ, dK), count_diff(r, r2, dI, dJ, dK));
    }
    // timing (inputs drift because sup/rhs are rewritten in place; diag-dominant keeps them finite)
    {
        dim3 grid((unsigned)cdiv(dI, 64), (unsigned)cdiv(dJ, 4));
        const double ms = time_ms([&](int) {
            hipLaunchKernelGGL((tridiag_generic_kernel<double>), grid, dim3(256), 0, 0, a.cview(), d.cview(), s2.view(), r2.view(), o2.view(), dI, dJ, dK);
        }, 3, 1);
        report("tridiag64", "1024x1024x160 generic", ms, (double)dI * dJ * dK, 56.0);
    }
    tridiag_variant<double, 2, 4>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_variant<double, 2, 2>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_variant<double, 2, 8>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_variant<double, 1, 4>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_variant<double, 1, 8>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_variant<double, 2, 1>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_variant<double, 1, 16>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_variant<double, 1, 8>(a, d, s, r, o, dI, dJ, dK, "1024x1024x160");
    tridiag_stack_variant<32, 40, 8>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<32, 32, 16>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<32, 48, 16>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<16, 48, 16>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<32, 48, 8>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<40, 40, 8>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<24, 40, 8>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<32, 36, 4>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<28, 40, 4>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<36, 40, 4>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
    tridiag_stack_variant<32, 40, 2>(a, d, s, r, o, s2, r2, o2, dI, dJ, dK);
}

// ---------------------------------------------------------------------------------------------
// Placement study for the tridiagonal solve: the five equally shaped fields live in ONE allocation at
// controlled relative offsets, field f at f * (field bytes rounded up to `round`) + f * delta.  Shows how the
// solve's speed depends on the fields' addresses relative to each other (the bimodal 84 / 95 GLUPS of round 1).
static void section_triplace(int argc_extra, const std::vector<std::string>& extra) {
    const int dI = 1024, dJ = 1024, dK = 160;
    const int64_t sj = dI, sk = (int64_t)dI * dJ;
    const size_t fbytes = (size_t)sk * dK * sizeof(double);
    const size_t MiB = 1 << 20;
    const size_t max_delta = 40 * MiB, round = 32 * MiB;
    const size_t slot = (fbytes + round - 1) / round * round;
    const size_t total = 5 * (slot + max_delta) + round;
    char* raw = nullptr;
    const bool contiguous = getenv("MB_CONTIG") && atoi(getenv("MB_CONTIG")) != 0;  // physically contiguous: offsets are PHYSICAL offsets
    if (contiguous) CK(hipExtMallocWithFlags((void**)&raw, total, hipDeviceMallocContiguous));
    else CK(hipMalloc(&raw, total));
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(raw) + round - 1) / round * round);
    printf("triplace   one allocation of %.1f MiB at %p (base rounded to 32 MiB: %p), field %.1f MiB, slot %.1f MiB\n",
           total / (double)MiB, (void*)raw, (void*)base, fbytes / (double)MiB, slot / (double)MiB);
    {  // where would five separate hipMallocs land?
        void* q[5];
        for (int f = 0; f < 5; ++f) CK(hipMalloc(&q[f], fbytes + 512));
        printf("triplace   five separate hipMalloc(%zu): ", fbytes + 512);
        for (int f = 0; f < 5; ++f) printf("%p%s", q[f], f < 4 ? " " : "\n");
        printf("triplace   differences to the first, in MiB:");
        for (int f = 1; f < 5; ++f) printf(" %.4f", ((char*)q[f] - (char*)q[0]) / (double)MiB);
        printf("\n");
        for (int f = 0; f < 5; ++f) hipFree(q[f]);
    }
    const size_t deltas_kib[] = {0,    4,    16,   64,   128,  256,  384,  512,  768,  1024, 1280, 1536, 1792,
                                 2048, 2560, 3072, 3584, 4096, 5120, 6144, 8192, 12288, 16384, 24576, 32768,
                                 1536 + 64, 1536 + 4, 1024 + 4, 2048 + 4, 2048 + 64, 4096 + 64};
    const unsigned ti = (unsigned)cdiv(dI, 64);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t dk_ : deltas_kib) {
            const size_t delta = dk_ * 1024;
            double* p[5];
            for (int f = 0; f < 5; ++f) p[f] = reinterpret_cast<double*>(base + f * (slot + delta));
            const size_t n = (size_t)sk * dK;
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[0], n, 1u, -1.0, 1.0);
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[1], n, 2u, 4.0, 5.0);
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[2], n, 3u, -1.0, 1.0);
            hipLaunchKernelGGL(fill_kernel<double>, dim3(4096), dim3(256), 0, 0, p[3], n, 4u, -10.0, 10.0);
            CK(hipDeviceSynchronize());
            View<const double> a{p[0], 1, sj, sk}, d{p[1], 1, sj, sk};
            View<double> s{p[2], 1, sj, sk}, r{p[3], 1, sj, sk}, o{p[4], 1, sj, sk};
            const double ms = time_ms([&](int) {
                if (contiguous) hipLaunchKernelGGL((tridiag_pipe_kernel<double, 104, 40, 4>), dim3(ti * dJ), dim3(64), 0, 0, a, d, s, r, o, dI, dJ, dK, ti);
                else hipLaunchKernelGGL((tridiag_stack_kernel<double, 32, 40, 8>), dim3(ti * dJ), dim3(64), 0, 0, a, d, s, r, o, dI, dJ, dK, ti);
            }, 5, 1);
            const double ms2 = time_ms([&](int) {
                const unsigned t2 = (unsigned)cdiv(dI, 256);
                hipLaunchKernelGGL((tridiag_kernel<double, 1, 8>), dim3(t2 * dJ), dim3(256), 0, 0, a, d, s, r, o, dI, dJ, dK, t2);
            }, 5, 1);
            char cfg[96];
            snprintf(cfg, sizeof cfg, "delta %6zu KiB  stack<32,40,8>", dk_);
            report("triplace", cfg, ms, (double)dI * dJ * dK, 56.0);
            snprintf(cfg, sizeof cfg, "delta %6zu KiB  two-sweep<1,8>", dk_);
            report("triplace", cfg, ms2, (double)dI * dJ * dK, 56.0);
        }
    hipFree(raw);
}

// ---------------------------------------------------------------------------------------------
// What does cross-stream synchronisation cost on the main stream?  Per "step": a 512x64x512 strip
// kernel (~46 us) and a 2-row kernel (~5 us), with (a) nothing, (b) hipEventRecord +
// hipStreamWaitEvent on an already-completed event of another stream, (c) a side stream doing a
// small kernel per step joined with events (the shape of the distributed step).
__global__ void tiny_kernel(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1u); }

static void section_events() {
    const int dI = 512, dJ = 64, dK = 512;
    DevField<double> in(dI, dJ, dK, 1, 1), out(dI, dJ, dK, 1, 1);
    fill(in, 1, -1.0, 1.0);
    unsigned* ctr;
    CK(hipMalloc(&ctr, 4));
    CK(hipMemset(ctr, 0, 4));
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ready, done;
    CK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    CK(hipEventRecord(done, side));
    const unsigned tx = 1, ty = dJ / 8, n = tx * ty * dK;
    auto big = [&]() { hipLaunchKernelGGL((lap5_strip_kernel<double, double, 0, 2, 8, 256, 4>), dim3(n), dim3(256), 0, 0, in.cview(), out.view(), dI, dJ, tx, ty); };
    auto rows = [&]() { hipLaunchKernelGGL((lap5_rows_kernel<double, double, 0, 2, 256>), dim3(2 * dK), dim3(256), 0, 0, in.cview(), out.view(), dI, 0, dJ - 1, 1u, 2u); };
    const int iters = 300;
    double ms;
    ms = time_ms([&](int) { big(); }, iters);
    printf("events     big kernel only                                  %8.1f us/step\n", ms * 1e3);
    ms = time_ms([&](int) { rows(); big(); }, iters);
    printf("events     rows + big (same stream, no events)              %8.1f us/step\n", ms * 1e3);
    ms = time_ms([&](int) { rows(); CK(hipEventRecord(ready, 0)); big(); }, iters);
    printf("events     rows + record + big                              %8.1f us/step\n", ms * 1e3);
    ms = time_ms([&](int) { CK(hipStreamWaitEvent(0, done, 0)); rows(); big(); }, iters);
    printf("events     wait(completed event) + rows + big               %8.1f us/step\n", ms * 1e3);
    ms = time_ms([&](int) { CK(hipStreamWaitEvent(0, done, 0)); rows(); CK(hipEventRecord(ready, 0)); big(); }, iters);
    printf("events     wait + rows + record + big                       %8.1f us/step\n", ms * 1e3);
    ms = time_ms([&](int) {
        CK(hipStreamWaitEvent(0, done, 0));
        rows();
        CK(hipEventRecord(ready, 0));
        CK(hipStreamWaitEvent(side, ready, 0));
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, ctr);
        big();
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, ctr);
        hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, ctr);
        CK(hipEventRecord(done, side));
    }, iters);
    printf("events     full choreography with 3 tiny side-stream kernels %7.1f us/step\n", ms * 1e3);
    CK(hipDeviceSynchronize());
    hipFree(ctr);
}

int main(int argc, char** argv) {
    std::vector<std::string> want;
    for (int i = 1; i < argc; ++i) want.push_back(argv[i]);
    auto on = [&](const char* s) {
        if (want.empty()) return true;
        for (auto& w : want)
            if (w == s) return true;
        return false;
    };
    char info[256];
    if (gt4mi_device_info(info, sizeof info) == 0) printf("%s\n", info);
    bool ok = true;
    if (on("dpp")) ok &= section_dpp();
    if (on("copy")) section_copy();
    if (!want.empty() && on("copynt")) section_copynt();
    if (!want.empty() && on("lapnt")) section_lapnt();
    if (!want.empty() && on("lap128")) section_lap128();
    if (!want.empty() && on("mix")) section_mix();
    if (on("lap")) section_lap();
    if (!want.empty() && on("lap512")) lap_suite(512, 512, 512, getenv("MB_LAP_EXTRA_PITCH") ? atoi(getenv("MB_LAP_EXTRA_PITCH")) : 0, "512^3");  // (-16: the 128-byte rows of round 4, pitch 528)
    if (!want.empty() && on("lapalign")) {
        // row alignment of the storage preset: 32 items (256 B, the gt:gpu value) vs 16 items (128 B = one L2
        // line: the east halo of a row and the west halo of the next then share a line)
        for (int align : {32, 16, 8}) {
            const int dI = 512, dJ = 512, dK = 512;
            DevField<double> in(dI, dJ, dK, 1, 1, align, 0), out(dI, dJ, dK, 1, 1, align, 0);
            fill(in, 1337, -1.0, 1.0);
            CK(hipMemset(out.raw, 0, out.bytes));
            const int64_t d[3] = {dI, dJ, dK};
            char cfg[64];
            for (int rep = 0; rep < 3; ++rep) {
                const double ms = time_ms([&](int) { lap5_launch_variant<double, double, 0>(in.cview(), out.view(), d, 0); }, 20);
                snprintf(cfg, sizeof cfg, "512^3 library default, rows aligned to %d items (pitch %lld)", align, (long long)in.sj);
                report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
            }
        }
    }
    if (!want.empty() && on("lapnarrow")) {
        // The 128-column share of the 4 x 2 grid (128 x 256 x 512) runs at 51 us where 512 x 64 x 512 -- the same number of points --
        // takes 46: the row pitch (130 items padded to 160: a fifth of every DRAM page is padding) or the kernel (one wave per row
        // strip, two edge lanes per wave)?  Rows aligned to 32 / 16 / 8 items, the launch the library would take, and 256-thread
        // workgroups of four strips (the shape of the one-launch step).
        for (int align : {32, 16, 8}) {
            for (auto dims : {std::array<int, 3>{128, 256, 512}, std::array<int, 3>{512, 64, 512}, std::array<int, 3>{256, 128, 512}}) {
                const int dI = dims[0], dJ = dims[1], dK = dims[2];
                DevField<double> in(dI, dJ, dK, 1, 1, align, 0), out(dI, dJ, dK, 1, 1, align, 0), in2(dI, dJ, dK, 1, 1, align, 0), out2(dI, dJ, dK, 1, 1, align, 0);
                fill(in, 1337, -1.0, 1.0);
                fill(in2, 1338, -1.0, 1.0);
                CK(hipMemset(out.raw, 0, out.bytes));
                CK(hipMemset(out2.raw, 0, out2.bytes));
                const int64_t d[3] = {dI, dJ, dK};
                char cfg[96];
                const double ms = time_ms([&](int i) {
                    if (i & 1) lap5_launch_variant<double, double, 0>(in2.cview(), out2.view(), d, 0);
                    else lap5_launch_variant<double, double, 0>(in.cview(), out.view(), d, 0);
                }, 40);
                snprintf(cfg, sizeof cfg, "%dx%dx%d library launch, rows aligned to %d items (pitch %lld)", dI, dJ, dK, align, (long long)in.sj);
                report("lap5_f64", cfg, ms, (double)dI * dJ * dK, 16.0);
            }
        }
    }
    if (!want.empty() && on("lapshare")) section_lapshare();
    if (!want.empty() && on("lappersist")) section_lappersist();
    if (on("hdiff")) section_hdiff();
    if (!want.empty() && on("hdiff2")) section_hdiff2();
    if (!want.empty() && on("hdiff3")) section_hdiff3();
    if (!want.empty() && on("hdiffnt")) section_hdiffnt();
    if (!want.empty() && on("hdiffl")) section_hdiffl();
    if (!want.empty() && on("kprobe")) section_kprobe();
    if (!want.empty() && on("hdiffxcd")) section_hdiffxcd();
    if (on("tridiag")) section_tridiag();
    if (!want.empty() && on("triplace")) section_triplace(0, want);
    if (!want.empty() && on("tripipe")) section_tripipe();
    if (!want.empty() && on("trint")) section_trint();
    if (!want.empty() && on("trimap")) section_trimap();
    if (!want.empty() && on("tripmc")) section_tripmc();
    if (!want.empty() && on("trilayout")) section_trilayout();
    if (!want.empty() && on("vadv")) section_vadv();
    if (!want.empty() && on("events")) section_events();
    return ok ? 0 : 1;
}

// device_info lives in gt4mi.hip; the micro-benchmark links stand-alone, so provide it here too.
extern "C" int gt4mi_device_info(char* buf, size_t buflen) {
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "device=%d name=%s arch=%s cus=%d clock_mhz=%d", dev, prop.name, prop.gcnArchName,
             prop.multiProcessorCount, prop.clockRate / 1000);
    return 0;
}
