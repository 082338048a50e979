// Shared helpers for the gfx950 stencil kernels (device + host launch side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "gt4py_amd.h"

// Bit-exact parity with the numpy backend forbids fused multiply-add contraction
// (SURVEY.md section 8a note N1).  The build also passes -ffp-contract=off; the pragma makes the
// guarantee local to the kernels in case a user rebuilds with other flags.
#pragma clang fp contract(off)

namespace gt4mi {

// ---- error reporting -----------------------------------------------------------------------
inline char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define GT4MI_HIP_CHECK(expr)                                                               \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::gt4mi::fail(GT4MI_ERR_HIP, "%s failed: %s (%s:%d)", #expr,             \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                \
    } while (0)

// ---- field views -----------------------------------------------------------------------------
// Pointer to the field's origin element plus element strides; what a kernel needs.
template <typename T>
struct View {
    T* p;            // element at the origin (first compute-domain point)
    int64_t si, sj, sk;  // strides in ELEMENTS
};

// Validate a field against a domain and a halo requirement, and build its origin-shifted view.
// halo_lo/hi: how far the stencil reaches below/above the compute domain along each axis.
template <typename T>
inline int make_view(const char* name, const gt4mi_field* f, const int64_t domain[3],
                     const int halo_lo[3], const int halo_hi[3], View<T>* v) {
    if (f == nullptr || f->data == nullptr)
        return fail(GT4MI_ERR_INVALID_ARGUMENT, "field '%s' is null", name);
    int64_t off = 0;
    for (int a = 0; a < 3; ++a) {
        if (f->stride[a] % (int64_t)sizeof(T) != 0)
            return fail(GT4MI_ERR_UNSUPPORTED,
                        "field '%s': byte stride %lld along axis %d is not a multiple of the item size",
                        name, (long long)f->stride[a], a);
        if (f->origin[a] < halo_lo[a])
            return fail(GT4MI_ERR_OUT_OF_BOUNDS,
                        "field '%s': origin %lld along axis %d too small, must be at least %d", name,
                        (long long)f->origin[a], a, halo_lo[a]);
        if (f->origin[a] + domain[a] + halo_hi[a] > f->shape[a])
            return fail(GT4MI_ERR_OUT_OF_BOUNDS,
                        "field '%s': shape %lld along axis %d too small for origin %lld + domain %lld + halo %d",
                        name, (long long)f->shape[a], a, (long long)f->origin[a],
                        (long long)domain[a], halo_hi[a]);
        off += f->origin[a] * f->stride[a];
    }
    v->p = reinterpret_cast<T*>(static_cast<char*>(f->data) + off);
    v->si = f->stride[0] / (int64_t)sizeof(T);
    v->sj = f->stride[1] / (int64_t)sizeof(T);
    v->sk = f->stride[2] / (int64_t)sizeof(T);
    return GT4MI_OK;
}

// ---- aliasing ---------------------------------------------------------------------------------
// The reference's numpy backend evaluates a statement's right-hand side completely before it assigns
// (/root/reference/src/gt4py/cartesian/gtc/numpy/npir_codegen.py:205-210), so a call whose output array is also
// an input is well defined there.  The kernels here read and write concurrently and promise the compiler
// `__restrict__` pointers; every entry point therefore compares the byte ranges its fields touch and either
// routes an alias whose result is the same in any evaluation order to an order-preserving kernel, or refuses
// the call (GT4MI_ERR_UNSUPPORTED) -- never a silently different answer.
struct ByteSpan {
    uintptr_t lo, hi;  // [lo, hi): smallest range that holds every element the kernel touches
};

template <typename V>
inline ByteSpan span_of(const V& v, const int64_t domain[3], const int halo_lo[3], const int halo_hi[3]) {
    const int64_t strides[3] = {v.si, v.sj, v.sk};
    int64_t lo = 0, hi = 0;  // element offsets relative to v.p
    for (int a = 0; a < 3; ++a) {
        const int64_t first = -(int64_t)halo_lo[a], last = domain[a] - 1 + halo_hi[a];
        const int64_t x = first * strides[a], y = last * strides[a];
        lo += x < y ? x : y;
        hi += x < y ? y : x;
    }
    const uintptr_t base = reinterpret_cast<uintptr_t>(v.p);
    return ByteSpan{base + (uintptr_t)(lo * (int64_t)sizeof(*v.p)), base + (uintptr_t)((hi + 1) * (int64_t)sizeof(*v.p))};
}

inline bool spans_overlap(const ByteSpan& a, const ByteSpan& b) { return a.lo < b.hi && b.lo < a.hi; }

// Two views whose byte ranges overlap may still touch disjoint ELEMENTS: interleaved slices of one buffer
// (vel[..., 0] / vel[..., 1]), the J halves of an I-contiguous parent.  Provable when both views have the same strides and
// those strides are nested (each larger than everything the smaller axes can add up to): an element of `a` at index x
// coincides with an element of `b` at index y iff  sum_axis (x - y) * stride == b.p - a.p, and with nested strides the
// differences n = x - y can be solved axis by axis from the largest stride down (two candidates per axis).
// Returns true only when NO pair of touched elements can coincide; false = cannot be shown (treat as overlapping).
template <typename VA, typename VB>
inline bool elements_disjoint(const VA& a, const int alo[3], const int ahi[3], const VB& b, const int blo[3], const int bhi[3],
                              const int64_t domain[3]) {
    if (sizeof(*a.p) != sizeof(*b.p) || a.si != b.si || a.sj != b.sj || a.sk != b.sk) return false;
    const int64_t bytes = (int64_t)(reinterpret_cast<uintptr_t>(b.p) - reinterpret_cast<uintptr_t>(a.p));
    if (bytes % (int64_t)sizeof(*a.p) != 0) return false;  // elements straddle each other
    const int64_t delta = bytes / (int64_t)sizeof(*a.p);
    int64_t s[3] = {a.si, a.sj, a.sk}, nlo[3], nhi[3];
    for (int x = 0; x < 3; ++x) {
        // x in [-alo, d-1+ahi], y in [-blo, d-1+bhi]  =>  n = x - y in [nlo, nhi]
        nlo[x] = -(int64_t)alo[x] - (domain[x] - 1 + bhi[x]);
        nhi[x] = domain[x] - 1 + ahi[x] + (int64_t)blo[x];
        if (s[x] < 0) { s[x] = -s[x]; const int64_t t = nlo[x]; nlo[x] = -nhi[x]; nhi[x] = -t; }
        if (s[x] == 0 && (nlo[x] != 0 || nhi[x] != 0)) return false;  // a broadcast axis: many indices, one element
    }
    int order[3] = {0, 1, 2};  // ascending stride
    for (int x = 0; x < 3; ++x)
        for (int y = x + 1; y < 3; ++y)
            if (s[order[y]] < s[order[x]]) { const int t = order[x]; order[x] = order[y]; order[y] = t; }
    int64_t below[3];  // what the axes with smaller strides can contribute at most (absolute value)
    int64_t acc = 0;
    for (int r = 0; r < 3; ++r) {
        const int x = order[r];
        below[r] = acc;
        if (r > 0 && s[x] != 0 && s[x] <= acc) return false;  // strides not nested: no unique decomposition
        const int64_t m = nhi[x] > -nlo[x] ? nhi[x] : -nlo[x];
        acc += m * s[x];
    }
    // depth-first over (at most) two candidates per axis, largest stride first
    struct Solver {
        const int64_t *s, *nlo, *nhi, *below;
        const int* order;
        bool hit(int r, int64_t rest) const {
            if (r < 0) return rest == 0;
            const int x = order[r];
            if (s[x] == 0) return hit(r - 1, rest);
            int64_t q = rest / s[x];
            if (rest % s[x] != 0 && rest < 0) --q;  // floor
            for (int64_t n = q; n <= q + 1; ++n) {
                if (n < nlo[x] || n > nhi[x]) continue;
                const int64_t left = rest - n * s[x];
                if ((left < 0 ? -left : left) > below[r]) continue;
                if (hit(r - 1, left)) return true;
            }
            return false;
        }
    } solver{s, nlo, nhi, below, order};
    return !solver.hit(2, delta);
}

// the views' touched elements (compute domain grown by the halos) may coincide somewhere
template <typename VA, typename VB>
inline bool views_overlap(const VA& a, const int alo[3], const int ahi[3], const VB& b, const int blo[3], const int bhi[3],
                          const int64_t domain[3]) {
    if (!spans_overlap(span_of(a, domain, alo, ahi), span_of(b, domain, blo, bhi))) return false;
    return !elements_disjoint(a, alo, ahi, b, blo, bhi, domain);
}

// the same elements, one to one (same origin element, same strides)
template <typename V1, typename V2>
inline bool same_view(const V1& a, const V2& b) {
    return (const void*)a.p == (const void*)b.p && a.si == b.si && a.sj == b.sj && a.sk == b.sk;
}

inline int check_domain(const int64_t domain[3]) {
    if (domain == nullptr) return fail(GT4MI_ERR_INVALID_ARGUMENT, "domain is null");
    for (int a = 0; a < 3; ++a) {
        if (domain[a] < 0 || domain[a] > INT32_MAX)
            return fail(GT4MI_ERR_INVALID_ARGUMENT, "invalid domain size %lld along axis %d",
                        (long long)domain[a], a);
    }
    return GT4MI_OK;
}

// True when a view can be accessed with VEC-wide vectors along I for every (j,k) row.
template <typename V>
inline bool vec_ok(const V& v, int vec) {
    if (v.si != 1) return false;
    if ((reinterpret_cast<uintptr_t>(v.p) % (vec * sizeof(*v.p))) != 0) return false;
    return (v.sj % vec == 0) && (v.sk % vec == 0);
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Dynamic LDS bytes the whole-domain kernels are launched with.  The kernels use no LDS; a workgroup that RESERVES some
// limits how many workgroups a CU holds (160 KB per CU), i.e. the waves and with them the bytes the kernel keeps in flight.
// An HBM-saturating stencil kernel at full occupancy has tens of MB outstanding, and every other kernel on the device then
// waits ~10 us per memory access -- the send/recv kernels of a halo exchange crawl (13 -> 170 us next to horizontal
// diffusion, profiles/r1_dist_hdiff_rehearsal.log).  The distributed steps throttle their INTERIOR kernel with this while
// an exchange is in flight next to it (gt4mi_dist_*); everything else launches with 0.
inline unsigned& launch_dynamic_lds() {
    static thread_local unsigned bytes = 0;
    return bytes;
}

struct ScopedLaunchLds {
    unsigned saved;
    explicit ScopedLaunchLds(unsigned bytes) : saved(launch_dynamic_lds()) { launch_dynamic_lds() = bytes; }
    ~ScopedLaunchLds() { launch_dynamic_lds() = saved; }
};

// bytes to reserve per workgroup so that at most `workgroups_per_cu` of them share a CU (0 = no limit)
inline unsigned lds_for_workgroups_per_cu(int workgroups_per_cu) {
    if (workgroups_per_cu <= 0) return 0;
    return (unsigned)((160 * 1024) / workgroups_per_cu) & ~255u;
}

inline int env_int(const char* name, int fallback) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : fallback;
}

// ---- device helpers ---------------------------------------------------------------------------
template <typename T, int N>
struct VecT;
template <>
struct VecT<double, 1> { using type = double; };
template <>
struct VecT<double, 2> { using type = double __attribute__((ext_vector_type(2))); };
template <>
struct VecT<float, 1> { using type = float; };
template <>
struct VecT<float, 2> { using type = float __attribute__((ext_vector_type(2))); };
template <>
struct VecT<float, 4> { using type = float __attribute__((ext_vector_type(4))); };

// Vector load/store of N contiguous elements into/from a register array.
template <typename T, int N>
__device__ __forceinline__ void vload(const T* p, T (&r)[N]) {
    if constexpr (N == 1) {
        r[0] = *p;
    } else {
        using V = typename VecT<T, N>::type;
        V v = *reinterpret_cast<const V*>(p);
#pragma unroll
        for (int e = 0; e < N; ++e) r[e] = v[e];
    }
}

template <typename T, int N, bool NT>
__device__ __forceinline__ void vstore(T* p, const T (&r)[N]) {
    if constexpr (N == 1) {
        if constexpr (NT) __builtin_nontemporal_store(r[0], p);
        else *p = r[0];
    } else {
        using V = typename VecT<T, N>::type;
        V v;
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = r[e];
        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
        else *reinterpret_cast<V*>(p) = v;
    }
}

// Workgroup id remap: the dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md,
// "Workgroup dispatch").  Give each XCD a contiguous range of logical tiles so that tiles that
// share halo rows also share an L2.  Performance only; any mapping is correct.
// Grouped variant: runs of G consecutive logical tiles go to one XCD, runs are dealt round-robin.
// Small G keeps the chip-wide access stream nearly sequential in memory (which the HBM channels
// like) while still letting G-1 of every G tile boundaries be shared inside one L2.
template <unsigned G>
__device__ __forceinline__ unsigned xcd_remap_grouped(unsigned b, unsigned n) {
    constexpr unsigned NX = 8;
    constexpr unsigned R = NX * G;  // tiles per round
    if (b >= (n / R) * R) return b;  // tail: identity
    const unsigned round = b / R, r = b % R;
    return round * R + (r % NX) * G + (r / NX);
}

__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
    constexpr unsigned NX = 8;
    const unsigned per = n / NX;  // tiles per XCD (full part)
    if (b >= per * NX) return b;  // tail: identity
    return (b % NX) * per + (b / NX);
}

}  // namespace gt4mi
