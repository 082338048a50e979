// The direct halo transport: peer stores from the pack kernel -- NEW (the reference is single-device; SURVEY.md section 5 names
// "direct peer stores from the pack kernel" as the alternative to RCCL point-to-point).
//
// Why: a halo message is <= 2 MB and an exchange per apply is latency bound.  With RCCL a face goes field -> send buffer (pack
// kernel) -> peer's receive buffer (the send/recv kernel: 13 us alone, 30-190 us next to an HBM-saturating interior kernel) ->
// ghost cells (unpack kernel).  Here the pack kernel stores every face straight into the NEIGHBOUR's receive buffer -- mapped
// into this process with hipIpcOpenMemHandle; over xGMI when the neighbour is another device -- and its last workgroup raises a
// flag at the neighbour; the neighbour's unpack kernel waits for its flags, copies, and raises the sender's "consumed" flag so
// that the buffer may be overwritten by the next exchange.  Two kernels per phase, no third party, everything stream ordered.
//
//   * ONE pool of fine-grained device memory per plan -- a page of flag words, then all receive buffers --, exported with
//     hipIpcGetMemHandle; a neighbour that is the rank itself (periodic axis of one rank) uses the pointer as it is;
//   * flags: 32-bit counters; "arrived" flags (one per receive) live in the RECEIVER's pool, "consumed" flags (one per send) in
//     the SENDER's: whoever waits polls its own memory, whoever signals stores over the link; both carry the number of the
//     exchange, so nothing is ever reset.  (Flags in registered host memory worked too and cost 300-700 us per exchange: every
//     polling workgroup is a PCIe read);
//   * ordering without fences: the writer's stores to the peer are write-through (sc0 sc1: system scope), every wave waits for
//     their acknowledgement (s_waitcnt vmcnt(0)), barrier, then one lane ADDS 1 to the flag (the flags count workgroups:
//     16 KB of a message each, on both sides); on the reader's side ONE lane polls the flag, a barrier, and only THEN the
//     workgroup loads its receive buffer, past the caches (sc0 sc1 loads): message passing needs the payload loads behind the
//     flag load.  (Rounds 2-3 issued the payload loads first and looked at the flag while they were in flight, reloading only
//     when the first look said "not yet": a payload load served before the peer's store landed and a flag load served after
//     its add -- different channels, no order between them -- delivered the PREVIOUS exchange's ghost values with a flag that
//     said "ready".  One dependent round trip cheaper, and wrong.)  Neither side ever writes back or invalidates a cache
//     (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 stores AND sc1 loads");
//   * a wait is bounded (GT4MI_PLAN_DIRECT_TIMEOUT_MS / GT4MI_DIRECT_TIMEOUT_MS, default 30 s) and a wait that runs out FAILS
//     HARD: the workgroup copies nothing and signals nothing (no garbage in ghost cells, no buffer overwritten that the peer has
//     not consumed), one lane sets the plan's error word -- host memory mapped into the device --, and every later call on the
//     plan (exchange, fused step, gt4mi_halo_exchange_end) reads that word without synchronising and returns
//     GT4MI_ERR_TIMEOUT.  The counts never match again: the plan stays failed until it is destroyed (on every rank);
//   * who talks to whom is set up by the host side (gt4py_amd/distributed/native.py: the k-th send to a peer lands in the buffer
//     of the k-th receive that peer posted for this rank, exactly RCCL's matching rule).
//
// Rehearsed on ONE device: every neighbour the rank itself, and TWO PROCESSES sharing the device (tests/test_gpu_distributed.py)
// -- real IPC mappings, real cross-process flags; what a 1-GPU box cannot show is the same stores crossing xGMI.
#pragma once

#include <unistd.h>

#include <cstring>

namespace gt4mi {

constexpr size_t DIRECT_FLAG_BYTES = 4096;  // the first page of a pool: the flag words (up to 1023 messages; the last word is scratch)

struct DirectBatch {
    uint32_t* wait_flag[BoxBatch::MAX];    // the copies of box m may be stored when *wait_flag[m] >= wait_value[m]
    uint32_t* signal_flag[BoxBatch::MAX];  // ... and every workgroup of box m adds 1 there when its part is done
    uint32_t wait_value[BoxBatch::MAX];
    unsigned blocks[BoxBatch::MAX];        // workgroups that work on box m (the others of the launch leave at once)
    uint32_t* error;                       // host memory mapped into the device: a wait ran out of time
    long long timeout_ticks;               // of the 100 MHz wall clock
    int fenced;                            // GT4MI_PLAN_DIRECT_FENCED (wave-uniform): see direct_release_fence / direct_acquire_fence
};

// FENCED MODE (GT4MI_PLAN_DIRECT_FENCED = 1; the fall-back between "direct" and RCCL).  The default ordering above rests on two
// properties of the memory system that one device can show and xGMI may not share: a write-through (sc0 sc1) store that has been
// acknowledged (vmcnt) is visible to every later load of every agent, and an sc0 sc1 load issued behind the flag load is served
// behind it.  Fenced mode asks the ISA's own memory model instead: the signalling side does a SYSTEM-scope release before its
// add (the compiler's `buffer_wbl2 sc0 sc1` + `s_waitcnt vmcnt(0) lgkmcnt(0)`: every store of the wave, and every dirty L2 line,
// is at its home before the flag moves), the waiting side a SYSTEM-scope acquire after its flag load (`buffer_inv sc0 sc1`:
// nothing cached or prefetched before the flag is used behind it).  Cost: the release next to an interior kernel that keeps the
// L2 full of dirty lines -- measured on the self-loop; that is why it is not the default.
__device__ __forceinline__ void direct_release_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
__device__ __forceinline__ void direct_acquire_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }

// One lane waits until *flag has reached `value` (counters wrap: "has reached" = the signed difference is not negative).  False
// = out of time: the plan's error word is set, the caller must neither copy nor signal.
__device__ __forceinline__ bool direct_wait(const uint32_t* flag, uint32_t value, long long timeout_ticks, uint32_t* error) {
    if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) >= 0) return true;
    const long long t0 = wall_clock64();  // 100 MHz
    int nap = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
        if (wall_clock64() - t0 > timeout_ticks) {  // the peer is not coming
            __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        // back off: hundreds of waves may poll ONE word (the edge units of a fused step whose face is late), and the sender's add has
        // to get through the same L2 channel
        if (nap < 4) __builtin_amdgcn_s_sleep(8);
        else if (nap < 16) __builtin_amdgcn_s_sleep(32);
        else __builtin_amdgcn_s_sleep(127);
        ++nap;
    }
    return true;
}

constexpr int DIRECT_UNROLL = 4;                                   // 16-byte vectors per thread
constexpr int64_t DIRECT_VECTORS_PER_BLOCK = 256 * DIRECT_UNROLL;  // 16 KB of every message per workgroup -- on BOTH sides: the
// flags count workgroups, and the receiver must know how many the sender's launch had without being told

inline unsigned direct_blocks(size_t message_bytes) { return (unsigned)((message_bytes / 16 + DIRECT_VECTORS_PER_BLOCK) / DIRECT_VECTORS_PER_BLOCK); }

// 16 bytes to the peer's receive buffer, written THROUGH the caches (sc0 sc1 = system scope): once the wave's vmcnt is 0 the data
// is where the peer will read it, and no cache holds a dirty copy that a fence would have to write back.
__device__ __forceinline__ void direct_store(u32x4* where, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(where), "v"(v) : "memory");
}
// 16 bytes from my receive buffer, past the caches (another agent wrote them; a cached copy would be the previous exchange's).
// The caller waits (s_waitcnt vmcnt(0)) before it uses the value: the compiler does not see this load.
__device__ __forceinline__ u32x4 direct_load(const u32x4* where) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(where) : "memory");
    return v;
}

// The copy of halo_batch_kernel (comm.hip.h) with a wait in front and a signal behind.  PACK: field box -> `buffer` (the peer's
// receive buffer; wait: the peer has unpacked the previous exchange); else `buffer` (my receive buffer) -> field box (wait: the
// data has arrived).  Next to an HBM-saturating interior kernel every DEPENDENT memory round trip costs ~10 us.  PACK has two
// of them: (flag poll || loads of my own field), then the stores; the signal is a posted add.  The unpack has three: the flag,
// THEN the loads of the receive buffer (see "ordering" above), then the stores.
template <typename U, bool PACK>
__device__ __forceinline__ void direct_block(U* field, int64_t si, int64_t sj, int64_t sk, const BoxBatch& b, const DirectBatch& d,
                                             const int m, const unsigned block) {
    constexpr int UNROLL = DIRECT_UNROLL;
    constexpr int PER_VEC = 16 / (int)sizeof(U);
    if (block >= d.blocks[m]) return;
    __shared__ int ready;
    const int ei_items = b.ext[m][0], ej = b.ext[m][1], ek = b.ext[m][2];
    const int64_t n = (int64_t)ei_items * ej * ek, nv = n / PER_VEC;  // items, whole 16-byte vectors of the dense buffer
    U* base = field + b.offset[m];
    u32x4* vbuf = static_cast<u32x4*>(b.buffer[m]);
    const bool rows_are_vectors = b.vec[m] != 0;
    const int ei_vec = ei_items / PER_VEC;
    auto item_at = [&](int64_t t) -> U* {  // item t of the dense buffer, in the field
        const int i = (int)(t % ei_items);
        const int64_t r = t / ei_items;
        return base + i * si + (r % ej) * sj + (r / ej) * sk;
    };
    auto vector_at = [&](int64_t tv) -> u32x4* {  // (rows_are_vectors) vector tv of the dense buffer, in the field
        const int i = (int)(tv % ei_vec);
        const int64_t r = tv / ei_vec;
        return reinterpret_cast<u32x4*>(base + (int64_t)i * PER_VEC + (r % ej) * sj + (r / ej) * sk);
    };
    union Vec { u32x4 v; U item[PER_VEC]; };
    Vec x[UNROLL];
    const int64_t tv0 = (int64_t)block * DIRECT_VECTORS_PER_BLOCK + threadIdx.x;
    auto load_all = [&]() {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t tv = tv0 + (int64_t)u * 256;
            if (tv >= nv) continue;
            if constexpr (PACK) {
                if (rows_are_vectors) x[u].v = *vector_at(tv);
                else {
#pragma unroll
                    for (int e = 0; e < PER_VEC; ++e) x[u].item[e] = *item_at(tv * PER_VEC + e);
                }
            } else {
                x[u].v = direct_load(vbuf + tv);
            }
        }
    };
#ifdef GT4MI_DIRECT_ROUND3_LOAD_ORDER
    // (evidence build only, `make r3order`: the receive side as rounds 2-3 had it -- payload loads first, one look at the flag
    // while they are in flight, a reload only if that look said "not yet"; profiles/r4_two_rank_direct_loop.log)
    __shared__ int first_look;
    load_all();
    if (threadIdx.x == 0) {
        first_look = (int)(__hip_atomic_load(d.wait_flag[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - d.wait_value[m]) >= 0;
        ready = direct_wait(d.wait_flag[m], d.wait_value[m], d.timeout_ticks, d.error) ? 1 : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!ready) return;
    if constexpr (!PACK)
        if (!first_look) load_all();
#else
    if constexpr (PACK) load_all();  // my own field: in flight while lane 0 looks at the flag
    if (threadIdx.x == 0) ready = direct_wait(d.wait_flag[m], d.wait_value[m], d.timeout_ticks, d.error) ? 1 : 0;
    __syncthreads();
    if (!ready) return;  // out of time: nothing is copied, nothing is signalled -- the plan has failed (direct_failed)
    // Behind the flag.  The hardware assumption of the default mode: VMEM instructions of a wave issue in order and the barrier
    // above orders every wave behind lane 0's flag load; the compiler must not move the payload loads up either (the asm loads
    // are `volatile` with a memory clobber; the signal fence pins everything else).  Fenced mode adds the ISA's acquire.
    __atomic_signal_fence(__ATOMIC_ACQUIRE);
    if (d.fenced) direct_acquire_fence();
    if constexpr (!PACK) load_all();  // what the peer stored before it raised the flag
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the asm loads of the unpack side are invisible to the compiler)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int64_t tv = tv0 + (int64_t)u * 256;
        if (tv >= nv) continue;
        if constexpr (PACK) {
            direct_store(vbuf + tv, x[u].v);
        } else if (rows_are_vectors) {
            *vector_at(tv) = x[u].v;
        } else {
#pragma unroll
            for (int e = 0; e < PER_VEC; ++e) *item_at(tv * PER_VEC + e) = x[u].item[e];
        }
    }
    if (block == 0 && threadIdx.x < n - nv * PER_VEC) {  // the few items behind the last whole vector
        U* buf = static_cast<U*>(b.buffer[m]);
        const int64_t t = nv * PER_VEC + threadIdx.x;
        if constexpr (PACK) __hip_atomic_store(buf + t, *item_at(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else *item_at(t) = __hip_atomic_load(buf + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // PACK: every wave waits until what it stored to the peer has been acknowledged (write-through stores: then it IS there);
    // unpack: its loads from the receive buffer completed above.  No fence: a system-scope `buffer_wbl2` next to an interior
    // kernel that keeps the L2 full of dirty lines cost 300-800 us per exchange in the first version of this kernel.
    if constexpr (PACK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PACK)
        if (d.fenced) direct_release_fence();  // every wave: its stores (and whatever else is dirty) are at their home
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(d.signal_flag[m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // posted
}

template <typename U, bool PACK>
__global__ void __launch_bounds__(256)
halo_direct_kernel(U* field, int64_t si, int64_t sj, int64_t sk, BoxBatch b, DirectBatch d) {
    direct_block<U, PACK>(field, si, sj, sk, b, d, (int)blockIdx.y, blockIdx.x);
}

// How long a wait may take, in ticks of the 100 MHz wall clock: the plan's option, else GT4MI_DIRECT_TIMEOUT_MS, else 30 s (a
// neighbour that compiles a kernel, pages in a library or sits in a debugger is seconds late and not broken; RCCL would wait
// for ever).
inline long long direct_timeout_ticks(const gt4mi_halo_plan* plan) {
    static const int env = env_int("GT4MI_DIRECT_TIMEOUT_MS", 30000);
    const long long ms = plan->direct.timeout_ms > 0 ? plan->direct.timeout_ms : (env > 0 ? env : 30000);
    return ms * 100000LL;
}

// The plan's direct transport has failed -- a wait ran out of time on the device (the error word is host memory: read here
// without synchronising anything), or an exchange was enqueued only in part -- and stays failed: every entry point that
// touches the exchange asks this first.
inline int direct_failed(const gt4mi_halo_plan* plan) {
    const auto& dx = plan->direct;
    if (!dx.prepared) return GT4MI_OK;
    if (dx.broken)
        return fail(GT4MI_ERR_HIP, "direct transport: an earlier exchange of this plan was enqueued only in part (%s); its flags no longer "
                                   "count what the neighbours count: destroy the plan on every rank", dx.broken);
    if (dx.error && __atomic_load_n(dx.error, __ATOMIC_RELAXED) != 0u)
        return fail(GT4MI_ERR_TIMEOUT, "direct transport: a wait for a neighbour ran out of time (%lld ms) in one of the %u exchanges this "
                                       "plan has started; the workgroup that waited copied and signalled nothing, the ghost cells of that "
                                       "exchange are incomplete and the plan stays failed: destroy it on every rank",
                    direct_timeout_ticks(plan) / 100000LL, dx.step);
    return GT4MI_OK;
}

inline int direct_index(const gt4mi_halo_plan* plan, bool is_send, int phase, int m) {
    const size_t nr0 = plan->recvs[0].size(), nr = nr0 + plan->recvs[1].size(), ns0 = plan->sends[0].size();
    return is_send ? (int)(nr + (phase ? ns0 : 0) + m) : (int)((phase ? nr0 : 0) + m);
}

// The boxes and flags of one phase's pack (PACK) or unpack launch; `blocks` = workgroups per box of the launch (0: nothing to do).
template <typename U, bool PACK>
inline int direct_batches(gt4mi_halo_plan* plan, const gt4mi_field* f, int phase, BoxBatch& b, DirectBatch& d, int64_t& blocks) {
    const auto& msgs = PACK ? plan->sends[phase] : plan->recvs[phase];
    blocks = 0;
    b.n = 0;
    d.fenced = 0;
    if (msgs.empty()) return GT4MI_OK;
    if ((int)msgs.size() > BoxBatch::MAX) return fail(GT4MI_ERR_UNSUPPORTED, "halo: more than %d boxes per phase", BoxBatch::MAX);
    auto& dx = plan->direct;
    b.n = (int)msgs.size();
    constexpr int64_t PER_VEC = 16 / (int64_t)sizeof(U);
    const bool field_vec = f->stride[0] == (int64_t)sizeof(U) && f->stride[1] % 16 == 0 && f->stride[2] % 16 == 0 &&
                           reinterpret_cast<uintptr_t>(f->data) % 16 == 0;
    for (int m = 0; m < b.n; ++m) {
        int64_t off = 0;
        for (int a = 0; a < 3; ++a) {
            if (msgs[m].lo[a] + msgs[m].ext[a] > f->shape[a])
                return fail(GT4MI_ERR_OUT_OF_BOUNDS, "halo: box [%lld, %lld) outside of axis %d (size %lld)",
                            (long long)msgs[m].lo[a], (long long)(msgs[m].lo[a] + msgs[m].ext[a]), a, (long long)f->shape[a]);
            if (f->stride[a] % (int64_t)sizeof(U) != 0) return fail(GT4MI_ERR_UNSUPPORTED, "halo: stride not a multiple of the item size");
            off += msgs[m].lo[a] * (f->stride[a] / (int64_t)sizeof(U));
            b.ext[m][a] = (int)msgs[m].ext[a];
        }
        b.offset[m] = off;
        void* buffer = PACK ? static_cast<void*>(dx.send_to[phase][m]) : msgs[m].buffer;
        uint32_t* signal = PACK ? dx.signal_arrived[phase][m] : dx.signal_consumed[phase][m];
        if (buffer == nullptr || signal == nullptr)
            return fail(GT4MI_ERR_INVALID_ARGUMENT, "halo (direct transport): message %d of phase %d was never connected to its peer", m, phase);
        b.buffer[m] = buffer;
        b.vec[m] = field_vec && msgs[m].lo[0] % PER_VEC == 0 && msgs[m].ext[0] % PER_VEC == 0 && reinterpret_cast<uintptr_t>(buffer) % 16 == 0;
        // PACK: wait until the peer has unpacked what the previous exchange put into this buffer; else: wait for the data.  The
        // flags count workgroups (every exchange adds direct_blocks(bytes) to both): nothing is ever reset.
        const unsigned nb = direct_blocks(msgs[m].bytes);
        d.wait_flag[m] = dx.flags + direct_index(plan, PACK, phase, m);
        d.wait_value[m] = (PACK ? dx.step - 1 : dx.step) * nb;
        // (GT4MI_DIRECT_TEST_LOSE_SIGNALS=1 when the plan was prepared, tests only: the pushes signal into an unused word -- what a
        // broken link looks like from the receiver's side: its waits run out of time.  =2: only while the plan is NOT in fenced
        // mode -- a transport whose default mode fails and whose fenced mode works, for the tests of the fall-back ladder)
        const bool lose = dx.lose_signals == 1 || (dx.lose_signals == 2 && !dx.fenced);
        d.signal_flag[m] = (PACK && lose) ? dx.flags + DIRECT_FLAG_BYTES / sizeof(uint32_t) - 1 : signal;
        d.blocks[m] = nb;
        blocks = nb > blocks ? nb : blocks;
    }
    d.error = dx.error;
    d.timeout_ticks = direct_timeout_ticks(plan);
    d.fenced = dx.fenced;
    return GT4MI_OK;
}

// The receive boxes of one phase for kernels that read the receive buffers themselves (lap5_edge.hip.h), on EITHER transport: the
// direct transport's own flags, or -- the plan exchanges through RCCL and the kernel runs behind the send/recv kernel in stream
// order -- a flag word that is always satisfied and a sink for the signals (plan->edge_words).
template <typename U>
inline int direct_batches_recv(gt4mi_halo_plan* plan, const gt4mi_field* f, int phase, BoxBatch& b, DirectBatch& d) {
    int64_t blocks = 0;
    if (plan->transport == GT4MI_TRANSPORT_DIRECT) return direct_batches<U, false>(plan, f, phase, b, d, blocks);
    const auto& msgs = plan->recvs[phase];
    b.n = (int)msgs.size();
    if (b.n > BoxBatch::MAX) return fail(GT4MI_ERR_UNSUPPORTED, "halo: more than %d boxes per phase", BoxBatch::MAX);
    constexpr int64_t PER_VEC = 16 / (int64_t)sizeof(U);
    const bool field_vec = f->stride[0] == (int64_t)sizeof(U) && f->stride[1] % 16 == 0 && f->stride[2] % 16 == 0 &&
                           reinterpret_cast<uintptr_t>(f->data) % 16 == 0;
    for (int m = 0; m < b.n; ++m) {
        int64_t off = 0;
        for (int a = 0; a < 3; ++a) {
            if (msgs[m].lo[a] + msgs[m].ext[a] > f->shape[a])
                return fail(GT4MI_ERR_OUT_OF_BOUNDS, "halo: box [%lld, %lld) outside of axis %d (size %lld)",
                            (long long)msgs[m].lo[a], (long long)(msgs[m].lo[a] + msgs[m].ext[a]), a, (long long)f->shape[a]);
            if (f->stride[a] % (int64_t)sizeof(U) != 0) return fail(GT4MI_ERR_UNSUPPORTED, "halo: stride not a multiple of the item size");
            off += msgs[m].lo[a] * (f->stride[a] / (int64_t)sizeof(U));
            b.ext[m][a] = (int)msgs[m].ext[a];
        }
        b.offset[m] = off;
        b.buffer[m] = msgs[m].buffer;
        b.vec[m] = field_vec && msgs[m].lo[0] % PER_VEC == 0 && msgs[m].ext[0] % PER_VEC == 0 && reinterpret_cast<uintptr_t>(msgs[m].buffer) % 16 == 0;
        d.wait_flag[m] = plan->edge_words;  // always 0 >= 0
        d.wait_value[m] = 0;
        d.signal_flag[m] = plan->edge_words + 1;
        d.blocks[m] = direct_blocks(msgs[m].bytes);
    }
    d.error = plan->edge_words + 2;
    d.timeout_ticks = direct_timeout_ticks(plan);
    d.fenced = 0;  // (RCCL delivered the buffers: stream order, the send/recv kernel's own fences)
    return GT4MI_OK;
}

template <typename U, bool PACK>
inline int direct_copy(gt4mi_halo_plan* plan, const gt4mi_field* f, int phase, hipStream_t s) {
    BoxBatch b;
    DirectBatch d;
    int64_t blocks = 0;
    if (int rc = direct_batches<U, PACK>(plan, f, phase, b, d, blocks)) return rc;
    if (blocks == 0) return GT4MI_OK;
    hipLaunchKernelGGL((halo_direct_kernel<U, PACK>), dim3((unsigned)blocks, (unsigned)b.n), dim3(256), 0, s,
                       static_cast<U*>(f->data), f->stride[0] / (int64_t)sizeof(U), f->stride[1] / (int64_t)sizeof(U),
                       f->stride[2] / (int64_t)sizeof(U), b, d);
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

inline int direct_push(gt4mi_halo_plan* plan, const gt4mi_field* field, int phase, hipStream_t s) {
    if (!plan->direct.prepared) return fail(GT4MI_ERR_INVALID_ARGUMENT, "halo: the direct transport was never prepared");
    return plan->elem_size == 8 ? direct_copy<uint64_t, true>(plan, field, phase, s) : direct_copy<uint32_t, true>(plan, field, phase, s);
}

inline int direct_unpack(gt4mi_halo_plan* plan, const gt4mi_field* field, int phase, hipStream_t s) {
    return plan->elem_size == 8 ? direct_copy<uint64_t, false>(plan, field, phase, s) : direct_copy<uint32_t, false>(plan, field, phase, s);
}

// ---- set-up ---------------------------------------------------------------------------------------------------------------
inline int direct_prepare(gt4mi_halo_plan* plan, gt4mi_direct_info* out) {
    auto& dx = plan->direct;
    if (!dx.prepared) {
        // ONE allocation other processes can map: the flag words, then all receive buffers (256-byte aligned slots) -- replacing
        // the per-message allocations of the plan.  Fine-grained device memory: flags and payload are written by another agent
        // while kernels of this one read them, so no cache may keep a copy.
        const size_t nflags = plan->recvs[0].size() + plan->recvs[1].size() + plan->sends[0].size() + plan->sends[1].size();
        if ((nflags + 1) * sizeof(uint32_t) > DIRECT_FLAG_BYTES) return fail(GT4MI_ERR_UNSUPPORTED, "direct transport: %d messages", (int)nflags);
        std::vector<size_t> offsets[2];
        size_t bytes = DIRECT_FLAG_BYTES;
        for (int p = 0; p < 2; ++p)
            for (auto& m : plan->recvs[p]) {
                offsets[p].push_back(bytes);
                bytes += (m.bytes + 255) / 256 * 256;
            }
        // every allocation first; the plan changes only once all of them have succeeded (a plan whose preparation failed
        // keeps its own receive buffers and is destroyed like any other)
        void *pool = nullptr, *error = nullptr, *counters = nullptr;
        auto give_up = [&](int rc) {
            if (pool) (void)hipFree(pool);
            if (error) (void)hipHostFree(error);
            if (counters) (void)hipFree(counters);
            (void)hipGetLastError();
            return rc;
        };
        if (hipExtMallocWithFlags(&pool, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            pool = nullptr;
            return give_up(fail(GT4MI_ERR_UNSUPPORTED, "direct transport: this runtime offers no fine-grained device memory (hipDeviceMallocFinegrained)"));
        }
        // the error word: HOST memory the device can write (a lane whose wait ran out of time stores there, over PCIe, once);
        // the host reads it without a synchronising call (direct_failed)
        if (hipHostMalloc(&error, 64, hipHostMallocMapped) != hipSuccess) {
            error = nullptr;
            return give_up(fail(GT4MI_ERR_HIP, "direct transport: hipHostMalloc of the error word failed"));
        }
        memset(error, 0, 64);
        if (hipMalloc(&counters, 4 * sizeof(unsigned)) != hipSuccess) {
            counters = nullptr;
            return give_up(fail(GT4MI_ERR_HIP, "direct transport: hipMalloc of the ring counters failed"));
        }
        if (hipMemset(pool, 0, DIRECT_FLAG_BYTES) != hipSuccess || hipMemset(counters, 0, 4 * sizeof(unsigned)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess)
            return give_up(fail(GT4MI_ERR_HIP, "direct transport: clearing the flag words failed"));
        dx.pool_bytes = bytes;
        dx.pool = static_cast<char*>(pool);
        dx.flags = reinterpret_cast<uint32_t*>(pool);
        dx.error = static_cast<uint32_t*>(error);
        dx.ring_counters = static_cast<unsigned*>(counters);
        for (int p = 0; p < 2; ++p) {
            dx.recv_offset[p] = offsets[p];
            for (size_t m = 0; m < plan->recvs[p].size(); ++m) {
                if (plan->recvs[p][m].buffer) (void)hipFree(plan->recvs[p][m].buffer);
                plan->recvs[p][m].buffer = dx.pool + dx.recv_offset[p][m];
            }
            dx.send_to[p].assign(plan->sends[p].size(), nullptr);
            dx.signal_arrived[p].assign(plan->sends[p].size(), nullptr);
            dx.signal_consumed[p].assign(plan->recvs[p].size(), nullptr);
        }
        dx.lose_signals = env_int("GT4MI_DIRECT_TEST_LOSE_SIGNALS", 0);
        dx.prepared = true;
    }
    if (out) {
        memset(out, 0, sizeof *out);
        hipIpcMemHandle_t h;
        GT4MI_HIP_CHECK(hipIpcGetMemHandle(&h, dx.pool));
        static_assert(sizeof h <= sizeof out->pool_handle, "hipIpcMemHandle_t grew");
        memcpy(out->pool_handle, &h, sizeof h);
        out->pool_bytes = (int64_t)dx.pool_bytes;
        out->flag_words = (int64_t)(DIRECT_FLAG_BYTES / sizeof(uint32_t));
        out->pid = (int32_t)getpid();
        int dev = 0;
        (void)hipGetDevice(&dev);
        out->device = dev;
    }
    return GT4MI_OK;
}

// The peer's pool in this process: this plan's own (peer == nullptr), or mapped once per distinct peer.
inline int direct_peer(gt4mi_halo_plan* plan, const gt4mi_direct_info* peer, char** pool) {
    auto& dx = plan->direct;
    if (peer == nullptr) {
        *pool = dx.pool;
        return GT4MI_OK;
    }
    const std::string key(peer->pool_handle, sizeof peer->pool_handle);
    for (auto& p : dx.peers)
        if (p.pool_key == key) {
            *pool = p.pool;
            return GT4MI_OK;
        }
    if (peer->pid == (int32_t)getpid())  // (two ranks in one process do not exist: hipIpcOpenMemHandle refuses the exporter's own handle)
        return fail(GT4MI_ERR_UNSUPPORTED, "direct transport: the peer is another plan of this process");
    gt4mi_halo_plan::Direct::Peer p;
    p.pool_key = key;
    hipIpcMemHandle_t h;
    memcpy(&h, peer->pool_handle, sizeof h);
    void* mapped = nullptr;
    if (hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return fail(GT4MI_ERR_HIP, "direct transport: hipIpcOpenMemHandle failed for the receive buffers of pid %d (device %d)",
                    (int)peer->pid, (int)peer->device);
    }
    p.pool = static_cast<char*>(mapped);
    p.opened_pool = true;
    dx.peers.push_back(p);
    *pool = p.pool;
    return GT4MI_OK;
}

inline int direct_connect(gt4mi_halo_plan* plan, int phase, int is_send, int index, const gt4mi_direct_info* peer,
                          int64_t peer_pool_offset, int peer_flag_index) {
    auto& dx = plan->direct;
    if (!dx.prepared) return fail(GT4MI_ERR_INVALID_ARGUMENT, "direct transport: prepare the plan first");
    if (phase < 0 || phase > 1) return fail(GT4MI_ERR_INVALID_ARGUMENT, "direct transport: phase %d", phase);
    const size_t n = is_send ? plan->sends[phase].size() : plan->recvs[phase].size();
    if (index < 0 || (size_t)index >= n) return fail(GT4MI_ERR_INVALID_ARGUMENT, "direct transport: message %d of %d", index, (int)n);
    char* pool = nullptr;
    if (int rc = direct_peer(plan, peer, &pool)) return rc;
    uint32_t* flags = reinterpret_cast<uint32_t*>(pool);
    if (peer_flag_index < 0 || (size_t)peer_flag_index >= DIRECT_FLAG_BYTES / sizeof(uint32_t))
        return fail(GT4MI_ERR_INVALID_ARGUMENT, "direct transport: flag %d", peer_flag_index);
    if (is_send) {
        const int64_t pool_bytes = peer ? peer->pool_bytes : (int64_t)dx.pool_bytes;
        if (peer_pool_offset < (int64_t)DIRECT_FLAG_BYTES || peer_pool_offset + (int64_t)plan->sends[phase][index].bytes > pool_bytes)
            return fail(GT4MI_ERR_OUT_OF_BOUNDS, "direct transport: a message of %lld bytes at offset %lld of a pool of %lld",
                        (long long)plan->sends[phase][index].bytes, (long long)peer_pool_offset, (long long)pool_bytes);
        dx.send_to[phase][index] = pool + peer_pool_offset;
        dx.signal_arrived[phase][index] = flags + peer_flag_index;
    } else {
        dx.signal_consumed[phase][index] = flags + peer_flag_index;
    }
    return GT4MI_OK;
}

// COLLECTIVE in effect: the pool is mapped by the neighbours, whose last unpack kernels post their "consumed" adds into it and
// whose pushes may still be on their way -- the caller frees it only after EVERY rank of the decomposition has finished its
// last exchange on the device (NativeHaloExchanger.close: device synchronise, then a round over the ranks' control channel;
// INTEGRATION.md "closing a plan").  This rank's own kernels are waited for here.
inline void direct_release(gt4mi_halo_plan* plan) {
    auto& dx = plan->direct;
    if (!dx.prepared) return;
    (void)hipDeviceSynchronize();
    for (auto& p : dx.peers)
        if (p.opened_pool) (void)hipIpcCloseMemHandle(p.pool);
    dx.peers.clear();
    if (dx.error) (void)hipHostFree(dx.error);
    if (dx.ring_counters) (void)hipFree(dx.ring_counters);
    for (int p = 0; p < 2; ++p)
        for (auto& m : plan->recvs[p]) m.buffer = nullptr;  // they lived in the pool
    if (dx.pool) (void)hipFree(dx.pool);
    dx = gt4mi_halo_plan::Direct();
}

}  // namespace gt4mi
