// RCCL halo exchange driven from native code -- NEW: gt4py.cartesian has no communication layer
// (SURVEY.md section 8e).  One process per GPU; the communicator is created from a 128-byte unique id
// that the host side distributes (torch.distributed / any launcher).
//
// Why native: at 8 GPUs a 512^3 fp64 Laplacian step is ~45 us of kernel time per GPU; issuing pack,
// 4 point-to-point operations, unpack and 3-5 kernel launches from Python costs several times that.
// Here one C call enqueues the whole step on two HIP streams:
//
//   comm stream : wait(main) -> pack faces -> ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd (I faces)
//                 -> unpack -> same for J faces (including fresh I-halo columns => corners) -> record
//   main stream : interior kernel ......................................... wait(comm) -> strips
//
// xGMI is point-to-point, messages are <= 2 MB: the exchange is latency-bound, every neighbour uses
// its own link, and there is no collective on the path.
//
// A second transport needs no RCCL at all: direct.hip.h (peer stores from the pack kernel into hipIpc-mapped receive buffers).
//
// librccl is resolved with dlopen at first use (the copy PyTorch already loaded is reused when
// present), so the stencil library itself loads on machines without RCCL.
#pragma once

#include <dlfcn.h>

#include <string>
#include <vector>

#include "common.hip.h"
#include "halo.hip.h"

namespace gt4mi {

// ---- minimal RCCL surface (names and ABI of <rccl/rccl.h>) ---------------------------------------
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
enum { RCCL_UINT8 = 1 };  // ncclUint8 / ncclChar = 1 in nccl.h's ncclDataType_t (ncclInt8 = 0)

struct RcclApi {
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(RcclComm, int*) = nullptr;      // optional: what the communicator itself reports
    int (*CommUserRank)(RcclComm, int*) = nullptr;
    int (*CommCuDevice)(RcclComm, int*) = nullptr;
    bool ok = false;
};

inline RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // already in the process (PyTorch)?
            if (h) break;
        }
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
        if (!h) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
        a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
        a.CommCuDevice = reinterpret_cast<decltype(a.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
        return a;
    }();
    return api;
}

#define GT4MI_RCCL_CHECK(expr)                                                                      \
    do {                                                                                            \
        int _r = (expr);                                                                            \
        if (_r != 0)                                                                                \
            return ::gt4mi::fail(GT4MI_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                     \
                                 ::gt4mi::rccl().GetErrorString ? ::gt4mi::rccl().GetErrorString(_r) \
                                                                : "rccl error",                    \
                                 __FILE__, __LINE__);                                               \
    } while (0)

}  // namespace gt4mi

// ---- opaque objects of the C ABI -----------------------------------------------------------------
struct gt4mi_comm {
    gt4mi::RcclComm comm = nullptr;
    int nranks = 0, rank = 0;
};

struct gt4mi_halo_plan {
    gt4mi_comm* comm = nullptr;
    int elem_size = 8;
    struct Msg {
        int peer;
        int64_t lo[3], ext[3];
        size_t bytes;
        void* buffer;
    };
    std::vector<Msg> sends[2], recvs[2];  // [phase]
    hipStream_t stream = nullptr;         // side stream the exchange runs on in the overlapped form
    hipEvent_t ready = nullptr, done = nullptr;
    bool forked = false;                  // `ready` already recorded by gt4mi_halo_exchange_fork
    bool primed = false;                  // the exchange of a stepper's first input was started (gt4mi_halo_exchange_begin)
    bool done_recorded = false;           // `done` has been recorded at least once: gt4mi_halo_exchange_end has something to wait for
    // how the fused distributed steps are scheduled (gt4mi_halo_plan_set_option); -1 = the entry point's own default
    int schedule = -1;            // GT4MI_SCHEDULE_JOIN / _CHAIN / _SWAP / _SWAP_PACKED
    int interior_wg_per_cu = -1;  // occupancy limit of the interior kernel while the exchange runs next to it (0 = none)
    int edge_columns = -1;        // width of the W / E boxes a fused hdiff step leaves to the ring kernel (-1: default)
    int defer_join = 0;           // chain schedule: leave the final join to gt4mi_halo_exchange_end (independent applies)
    // concurrency probe (ensure_concurrent_stream)
    unsigned* probe = nullptr;            // two device words: flag, result
    // device words for kernels that read the receive buffers of an RCCL plan themselves (lap5_edge.hip.h): [0] a flag that is
    // always satisfied (stays 0), [1] a sink for signals, [2] a sink for the error word, [4 .. 7] counters of the four sides
    uint32_t* edge_words = nullptr;
    hipStream_t probed_main = nullptr;
    bool probed = false, concurrent = false;
    // ---- the direct transport (direct.hip.h): peer stores from the pack kernel instead of RCCL send/recv ----
    int transport = 0;  // GT4MI_TRANSPORT_RCCL / GT4MI_TRANSPORT_DIRECT
    struct Direct {
        bool prepared = false;
        char* pool = nullptr;               // ALL receive buffers of the plan, one allocation other processes can map (hipIpc)
        size_t pool_bytes = 0;
        std::vector<size_t> recv_offset[2];
        uint32_t* flags = nullptr;          // my flag words = the first page of the pool: [arrived: one per receive][consumed: one per send]
        uint32_t* error = nullptr;          // HOST memory mapped into the device: a wait ran out of time (direct_failed reads it)
        unsigned* ring_counters = nullptr;  // device: workgroups of a fused unpack + ring launch that have read their face (2 words)
        int timeout_ms = 0;                 // GT4MI_PLAN_DIRECT_TIMEOUT_MS (0: GT4MI_DIRECT_TIMEOUT_MS, else 30 s)
        int fenced = 0;                     // GT4MI_PLAN_DIRECT_FENCED: release / acquire fences around the flags (direct.hip.h "fenced mode")
        const char* broken = nullptr;       // an exchange was enqueued only in part: which step failed (the plan stays failed)
        int lose_signals = 0;               // tests: GT4MI_DIRECT_TEST_LOSE_SIGNALS when the plan was prepared (1: always, 2: only while the plan is unfenced)
        uint32_t step = 0;                  // exchanges started
        bool first_pushed = false;          // halo_pack_first already pushed the first phase of exchange `step`
        struct Peer {
            std::string pool_key;
            char* pool = nullptr;
            bool opened_pool = false;
        };
        std::vector<Peer> peers;
        std::vector<char*> send_to[2];              // per send: where the message lands (in the peer's pool)
        std::vector<uint32_t*> signal_arrived[2];   // per send: the peer's flag that says "it is there"
        std::vector<uint32_t*> signal_consumed[2];  // per receive: the SENDER's flag that says "unpacked, the buffer is free again"
    } direct;
};

namespace gt4mi {

// HIP multiplexes streams onto a handful of hardware queues (4 by default) and two streams that
// land on the same queue run strictly in submission order -- measured on MI355X: with the side
// stream on the main stream's queue the "overlapped" exchange simply ran after the interior kernel
// (97 us per 512x64x512 step instead of ~60).  There is no API to ask for a distinct queue, so the
// plan PROBES: a one-thread kernel on the main stream waits (bounded) for a flag that a one-thread
// kernel submitted AFTERWARDS on the side stream raises.  The flag only arrives if the two streams
// really execute concurrently; otherwise the side stream is replaced (colliding ones are kept alive
// until a good one is found, so that their queue is not handed out again).
__global__ void probe_wait_kernel(unsigned* words, long long timeout_ticks) {
    const long long t0 = wall_clock64();  // 100 MHz
    while (__hip_atomic_load(&words[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        if (wall_clock64() - t0 > timeout_ticks) {
            words[1] = 0u;
            return;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    words[1] = 1u;
}
__global__ void probe_set_kernel(unsigned* words) {
    __hip_atomic_store(&words[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline int ensure_concurrent_stream(gt4mi_halo_plan* plan, hipStream_t main_stream) {
    if (plan->probed && plan->probed_main == main_stream) return GT4MI_OK;
    std::vector<hipStream_t> colliding;
    bool ok = false;
    for (int attempt = 0; attempt < 12 && !ok; ++attempt) {
        GT4MI_HIP_CHECK(hipMemsetAsync(plan->probe, 0, 8, main_stream));
        GT4MI_HIP_CHECK(hipStreamSynchronize(main_stream));
        hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, main_stream, plan->probe, 30000LL /* 0.3 ms */);
        hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, plan->stream, plan->probe);
        GT4MI_HIP_CHECK(hipStreamSynchronize(main_stream));
        GT4MI_HIP_CHECK(hipStreamSynchronize(plan->stream));
        unsigned words[2] = {0, 0};
        GT4MI_HIP_CHECK(hipMemcpy(words, plan->probe, 8, hipMemcpyDeviceToHost));
        ok = words[1] == 1u;
        if (!ok) {
            colliding.push_back(plan->stream);
            GT4MI_HIP_CHECK(hipStreamCreateWithFlags(&plan->stream, hipStreamNonBlocking));
        }
    }
    for (hipStream_t s : colliding) (void)hipStreamDestroy(s);
    plan->probed = true;
    plan->probed_main = main_stream;
    plan->concurrent = ok;
    return GT4MI_OK;
}

// All boxes of one phase are packed (or unpacked) by ONE launch: blockIdx.y selects the box.  The
// exchange is latency-bound, so launches on its critical path are what matters.
struct BoxBatch {
    static constexpr int MAX = 8;  // a single-phase plan has up to 8 neighbours (4 faces + 4 corners)
    int n;
    int64_t offset[MAX];  // element offset of the box start inside the field
    int ext[MAX][3];
    int vec[MAX];         // 1: rows along I are whole 16-byte vectors in the field and in the buffer
    void* buffer[MAX];
};

// One box per blockIdx.y.  Boxes whose rows along I start and end on 16-byte boundaries (in the field and in the dense
// buffer) move as 16-byte vectors, UNROLL of them per thread with all loads issued first: next to an HBM-saturating
// interior kernel a copy is bound by the bytes it keeps in flight -- the one-item-per-thread form took 29 us for the two
// 2 MB faces of the 512 x 64 x 512 share where it takes 5 us alone (profiles/r3_dist_lap5_skewed_timeline.txt).  Other
// boxes (one-column faces of 8-byte items, odd origins) go item by item.
template <typename U, bool PACK>
__global__ void __launch_bounds__(256)
halo_batch_kernel(U* field, int64_t si, int64_t sj, int64_t sk, BoxBatch b) {
    constexpr int UNROLL = 4;
    constexpr int PER_VEC = 16 / (int)sizeof(U);
    const int m = blockIdx.y;
    const int ej = b.ext[m][1], ek = b.ext[m][2];
    if (b.vec[m]) {
        const int ei = b.ext[m][0] / PER_VEC;  // vectors per row
        const int64_t n = (int64_t)ei * ej * ek;
        U* base = field + b.offset[m];
        u32x4* buf = static_cast<u32x4*>(b.buffer[m]);
        for (int64_t t0 = ((int64_t)blockIdx.x * UNROLL) * 256 + threadIdx.x; t0 < n; t0 += (int64_t)gridDim.x * 256 * UNROLL) {
            u32x4 v[UNROLL];
            u32x4* where[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t t = t0 + (int64_t)u * 256;
                const int64_t tt = t < n ? t : n - 1;
                const int i = (int)(tt % ei);
                const int64_t r = tt / ei;
                where[u] = reinterpret_cast<u32x4*>(base + (int64_t)i * PER_VEC + (r % ej) * sj + (r / ej) * sk);
                v[u] = PACK ? *where[u] : buf[tt];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t t = t0 + (int64_t)u * 256;
                if (t < n) {
                    if constexpr (PACK) buf[t] = v[u];
                    else *where[u] = v[u];
                }
            }
        }
        return;
    }
    const int ei = b.ext[m][0];
    const int64_t n = (int64_t)ei * ej * ek;
    U* base = field + b.offset[m];
    U* buf = static_cast<U*>(b.buffer[m]);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int i = (int)(t % ei);
        const int64_t r = t / ei;
        const int j = (int)(r % ej);
        const int k = (int)(r / ej);
        U* f = base + i * si + j * sj + k * sk;
        if constexpr (PACK) buf[t] = *f;
        else *f = buf[t];
    }
}

template <typename U, bool PACK>
inline int plan_copy_batch(const gt4mi_field* f, const std::vector<gt4mi_halo_plan::Msg>& msgs, hipStream_t s) {
    if (msgs.empty()) return GT4MI_OK;
    if ((int)msgs.size() > BoxBatch::MAX) return fail(GT4MI_ERR_UNSUPPORTED, "halo: more than %d boxes per phase", BoxBatch::MAX);
    BoxBatch b;
    b.n = (int)msgs.size();
    constexpr int64_t PER_VEC = 16 / (int64_t)sizeof(U);
    const bool field_vec = f->stride[0] == (int64_t)sizeof(U) && f->stride[1] % 16 == 0 && f->stride[2] % 16 == 0 &&
                           reinterpret_cast<uintptr_t>(f->data) % 16 == 0;
    int64_t blocks = 0;
    for (int m = 0; m < b.n; ++m) {
        int64_t off = 0, n = 1;
        for (int a = 0; a < 3; ++a) {
            if (msgs[m].lo[a] + msgs[m].ext[a] > f->shape[a])
                return fail(GT4MI_ERR_OUT_OF_BOUNDS, "halo: box [%lld, %lld) outside of axis %d (size %lld)",
                            (long long)msgs[m].lo[a], (long long)(msgs[m].lo[a] + msgs[m].ext[a]), a, (long long)f->shape[a]);
            if (f->stride[a] % (int64_t)sizeof(U) != 0) return fail(GT4MI_ERR_UNSUPPORTED, "halo: stride not a multiple of the item size");
            off += msgs[m].lo[a] * (f->stride[a] / (int64_t)sizeof(U));
            b.ext[m][a] = (int)msgs[m].ext[a];
            n *= msgs[m].ext[a];
        }
        b.offset[m] = off;
        b.buffer[m] = msgs[m].buffer;
        b.vec[m] = field_vec && msgs[m].lo[0] % PER_VEC == 0 && msgs[m].ext[0] % PER_VEC == 0 &&
                   reinterpret_cast<uintptr_t>(msgs[m].buffer) % 16 == 0;
        const int64_t need = b.vec[m] ? cdiv(n / PER_VEC, 256 * 4) : cdiv(n, 256);
        blocks = need > blocks ? need : blocks;
    }
    if (blocks == 0) return GT4MI_OK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((halo_batch_kernel<U, PACK>), dim3((unsigned)blocks, (unsigned)b.n), dim3(256), 0, s,
                       static_cast<U*>(f->data), f->stride[0] / (int64_t)sizeof(U), f->stride[1] / (int64_t)sizeof(U),
                       f->stride[2] / (int64_t)sizeof(U), b);
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

template <bool PACK>
inline int plan_copy(const gt4mi_halo_plan* plan, const gt4mi_field* f, const std::vector<gt4mi_halo_plan::Msg>& msgs,
                     hipStream_t s) {
    if (plan->elem_size == 8) return plan_copy_batch<uint64_t, PACK>(f, msgs, s);
    return plan_copy_batch<uint32_t, PACK>(f, msgs, s);
}

inline int first_phase(const gt4mi_halo_plan* plan) {
    for (int phase = 0; phase < 2; ++phase)
        if (!plan->sends[phase].empty() || !plan->recvs[phase].empty()) return phase;
    return 2;
}

// Pack the faces of the first non-empty phase only (they depend on nothing but the field itself,
// so a caller can enqueue this ahead of its interior kernel).
inline int direct_push(gt4mi_halo_plan* plan, const gt4mi_field* field, int phase, hipStream_t s);    // direct.hip.h
inline int direct_unpack(gt4mi_halo_plan* plan, const gt4mi_field* field, int phase, hipStream_t s);
inline int direct_failed(const gt4mi_halo_plan* plan);

inline int halo_pack_first(gt4mi_halo_plan* plan, const gt4mi_field* field, hipStream_t s) {
    const int p = first_phase(plan);
    if (p > 1) return GT4MI_OK;
    if (plan->transport == GT4MI_TRANSPORT_DIRECT) {  // the pack IS the transfer
        if (int rc = direct_failed(plan)) return rc;
        ++plan->direct.step;  // (the push reads it)
        if (int rc = direct_push(plan, field, p, s)) {
            --plan->direct.step;  // nothing was launched: the counters still agree with the neighbours'
            return rc;
        }
        plan->direct.first_pushed = true;
        return GT4MI_OK;
    }
    return plan_copy<true>(plan, field, plan->sends[p], s);
}

// Enqueue the two-phase exchange of `field`'s ghost cells on stream `s`.
// `skip_last_unpack`: the caller's next kernel reads the receive buffers of the LAST non-empty phase itself (lap5_edge.hip.h:
// waits for the direct transport's flags there, or simply runs behind the send/recv kernel) -- no unpack launch for that phase.
inline int halo_exchange_on(gt4mi_halo_plan* plan, const gt4mi_field* field, hipStream_t s,
                            bool first_pack_done = false, bool skip_last_unpack = false) {
    int last_phase = -1;
    for (int phase = 0; phase < 2; ++phase)
        if (!plan->sends[phase].empty() || !plan->recvs[phase].empty()) last_phase = phase;
    const int p0 = first_phase(plan);
    if (plan->transport == GT4MI_TRANSPORT_DIRECT) {
        // every face is stored straight into its neighbour's receive buffer by the pack kernel, whose last workgroup raises the
        // neighbour's flag; the unpack kernel waits for its own flags, copies, and tells the senders that their buffers are free
        if (int rc = direct_failed(plan)) return rc;
        const bool pushed = first_pack_done && plan->direct.first_pushed;
        if (!pushed) ++plan->direct.step;
        bool launched = pushed;  // something of this exchange is already on the device
        for (int phase = 0; phase < 2; ++phase) {
            if (plan->sends[phase].empty() && plan->recvs[phase].empty()) continue;
            int rc = GT4MI_OK;
            if (!(pushed && phase == p0)) rc = direct_push(plan, field, phase, s);
            if (rc == GT4MI_OK) {
                launched = launched || !plan->sends[phase].empty();
                if (!(skip_last_unpack && phase == last_phase)) rc = direct_unpack(plan, field, phase, s);
            }
            if (rc != GT4MI_OK) {
                // nothing launched yet: as if the call had never been made; else the neighbours will count an exchange that this
                // rank never completes -- the plan has failed (the error of THIS call is the launch's own)
                if (!launched) --plan->direct.step;
                else plan->direct.broken = "a push or unpack launch of a later phase was refused";
                return rc;
            }
            launched = true;
        }
        plan->direct.first_pushed = false;
        return GT4MI_OK;
    }
    if (plan->comm->comm == nullptr)
        return fail(GT4MI_ERR_UNSUPPORTED, "halo: this communicator has no RCCL behind it (gt4mi_comm_create_local): switch the plan to the "
                                          "direct transport first");
    RcclApi& api = rccl();
    for (int phase = 0; phase < 2; ++phase) {
        auto& sends = plan->sends[phase];
        auto& recvs = plan->recvs[phase];
        if (sends.empty() && recvs.empty()) continue;
        if (!(first_pack_done && phase == p0))
            if (int rc = plan_copy<true>(plan, field, sends, s)) return rc;
        GT4MI_RCCL_CHECK(api.GroupStart());
        for (auto& m : sends) GT4MI_RCCL_CHECK(api.Send(m.buffer, m.bytes, RCCL_UINT8, m.peer, plan->comm->comm, s));
        for (auto& m : recvs) GT4MI_RCCL_CHECK(api.Recv(m.buffer, m.bytes, RCCL_UINT8, m.peer, plan->comm->comm, s));
        GT4MI_RCCL_CHECK(api.GroupEnd());
        if (skip_last_unpack && phase == last_phase) continue;
        if (int rc = plan_copy<false>(plan, field, recvs, s)) return rc;
    }
    return GT4MI_OK;
}

}  // namespace gt4mi

#include "direct.hip.h"
