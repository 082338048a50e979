// The rest of a distributed 5-point step -- unpack AND ring -- as wave-sized units that read the receive buffers themselves,
// for every process grid (round 4; lap5_ring_unpack.hip.h of round 3 did this for the S / N faces of 1 x N grids only).
//
// Round 3's step on a P x Q grid was push + interior, unpack, ring: the unpack (receive buffers -> ghost cells, 9-12 us) and a
// ring whose W / E boxes were 8-16 columns wide so that the interior kernel kept its 16-byte alignment (22 us for the
// 128 x 256 x 512 share of 4 x 2: 2.1 M points in 64-byte pieces) together cost more than half an interior kernel.  Here:
//
//   * the interior kernel covers ALL columns of the rows that read no S / N ghost row, on whole 16-byte lanes, and simply does
//     not STORE its first / last lane where there is a W / E neighbour (lap5_strip_lane<MASKED>: what it computes there from a
//     ghost value that has not arrived is dropped);
//   * a COLUMN unit -- a wave, one row per lane, EDGE_LEVELS levels -- takes 64 rows of the W (E) face: the ghost values from the dense
//     receive buffer (coalesced), its row's first (last) 16-byte lane and the column beyond it, the rows above and below from
//     the neighbouring lanes (DPP); it computes that whole lane (the interior stores nothing there) and writes the ghost cells
//     (the exchange's contract);
//   * a ROW unit takes 64 x VEC columns of the S (N) row as before; where the row ends at a W / E face its end lane takes that
//     ghost value from the W / E buffer too;
//   * whatever else the plan receives (the corner boxes of a single-phase table) is copied by direct_block.
//
// Every unit waits for the arrival flag of each buffer it reads (the wave polls one word; the loads of the buffer come BEHIND the
// flag: direct.hip.h "ordering"), the units of a workgroup count themselves out per buffer together, and the last readers tell the
// sender that the buffer is free.  On a
// plan that exchanges through RCCL the same units run behind the send/recv kernel in stream order: their flags are words that
// are always satisfied, their signals go nowhere.
//
// The units run as a kernel of their own (lap5_edge_kernel: behind the send/recv kernel, or behind the interior kernel of a
// two-stream schedule) or as workgroups of the ONE launch of the one-stream schedule on the direct transport (lap5_step_kernel:
// push | interior | edge units behind 85 % of the interior), which start while the interior's last strips drain and find their
// faces long arrived.  Same expression (lap5_expr) on the same values as the whole-domain kernel: bit-identical.
#pragma once

#include "direct.hip.h"
#include "lap5.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

// K levels per unit.  A unit is a serial chain -- flag, loads, stores, and on gfx9 the NEXT loads wait behind the previous stores
// (one counter, vmcnt, returns in order): with 8 levels in four chunks the 640 units of the 128 x 256 x 512 share took ~25 us
// of a 50 us interior.  Two levels, one chunk: four times the waves, a quarter of the chain; what made round 3 choose 8 -- a
// same-address atomic per unit -- is gone: the units of a workgroup count themselves out together (edge_block_done).
constexpr int EDGE_LEVELS = 2;

struct EdgeFaces {
    // side 0 = W, 1 = E (column units), 2 = S, 3 = N (row units)
    int have[4];                // a face of this side is received
    const void* buffer[4];      // dense: W / E [k][j] one item each; S / N [k][i]
    int lo[4], ext[4];          // along the face, relative to the domain: first row (W / E) or column (S / N), and length
    uint32_t* wait_flag[4];
    uint32_t wait_value[4];
    uint32_t* consumed_flag[4];  // at the sender
    uint32_t consumed_add[4];
    unsigned* counter[4];        // units of this launch that have read the buffer
    unsigned readers[4];
    unsigned first[5];           // unit ranges: [first[s], first[s + 1]) are the units of side s
    unsigned tiles[4];           // units per level group of side s (tiles along the face)
    int rows_lo, rows_hi;        // column units COMPUTE rows [rows_lo, rows_hi) (the row units own the first / last row)
    uint32_t* error;
    long long timeout_ticks;
    int fenced;                  // GT4MI_PLAN_DIRECT_FENCED: an acquire behind the flags (direct.hip.h "fenced mode")
};

// One wave: has *flag reached value?  Every lane polls the same word (one request); false = out of time (error word set).
__device__ __forceinline__ bool edge_wait(const uint32_t* flag, uint32_t value, long long timeout_ticks, uint32_t* error) {
    return direct_wait(flag, value, timeout_ticks, error);
}

// The four waves of a workgroup have read the buffers in `mask` (bit s: this wave read side s's buffer; all loads returned):
// count them out together -- one atomic per side and workgroup --; the last readers of the launch free the buffer at the sender.
// EVERY wave of the workgroup calls this (no early return in front of it).
__device__ __forceinline__ void edge_block_done(const EdgeFaces& g, unsigned mask) {
    __shared__ unsigned read_by[4];
    if (threadIdx.x < 4) read_by[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int s = 0; s < 4; ++s)
            if (mask >> s & 1u) atomicAdd(&read_by[s], 1u);
    __syncthreads();
    if (threadIdx.x < 4 && read_by[threadIdx.x]) {
        const int s = (int)threadIdx.x;
        const unsigned n = read_by[s], before = atomicAdd(g.counter[s], n);
        if (before + n == g.readers[s]) {  // the launch's last readers of this buffer
            atomicExch(g.counter[s], 0u);  // (the next launch on this plan starts after this one: stream order)
            __hip_atomic_fetch_add(g.consumed_flag[s], g.consumed_add[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename U>
__device__ __forceinline__ U edge_load(const U* p) {  // past the caches: another agent wrote it
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// 64 x VEC columns of the S (f = 2) or N (f = 3) row of EDGE_LEVELS levels, CH levels at a time (all loads of a chunk -- the
// field's rows and the ghost values -- in flight together).  Two levels keep the unit inside the register budget of the
// interior's strips when both share a kernel: with 8 levels at once lap5_step_kernel needed 184 registers, 2 waves per SIMD
// instead of 6, and its interior part ran at half speed.
// (F, the side, is a template argument: an index into the by-value EdgeFaces that is not a constant makes the compiler keep a
// copy of those arrays in scratch memory)
template <typename T, typename W, int VARIANT, int VEC, int CH, int F>
__device__ __forceinline__ unsigned lap5_edge_row_unit(const View<T>& in, const View<T>& out, int dI, int dJ, int dK, const EdgeFaces& g,
                                                       const unsigned unit) {
    constexpr int f = F;
    using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    constexpr int LG = EDGE_LEVELS;
    static_assert(LG % CH == 0, "chunks of CH levels");
    const unsigned tiles_x = g.tiles[f];
    const int k0 = (int)(unit / tiles_x) * LG;
    const unsigned tx = unit % tiles_x;
    const int lane = (int)(threadIdx.x & 63);
    int i0 = (int)(tx * 64 + lane) * VEC;
    const bool active = i0 < dI;
    if (i0 >= dI) i0 = dI - VEC;
    const int j = f == 2 ? 0 : dJ - 1, gj = f == 2 ? -1 : dJ, far = f == 2 ? 1 : dJ - 2;
    const bool first = active && i0 == 0, last = active && i0 + VEC == dI;
    const bool corner_w = first && g.lo[f] < 0, corner_e = last && g.lo[f] + g.ext[f] > dI;  // (a face that carries the corner columns)
    const bool need_w = tx == 0 && g.have[0], need_e = tx == tiles_x - 1 && g.have[1];        // wave-uniform
    bool ready = edge_wait(g.wait_flag[f], g.wait_value[f], g.timeout_ticks, g.error);
    if (need_w) ready = edge_wait(g.wait_flag[0], g.wait_value[0], g.timeout_ticks, g.error) && ready;
    if (need_e) ready = edge_wait(g.wait_flag[1], g.wait_value[1], g.timeout_ticks, g.error) && ready;
    if (!ready) return 0u;  // out of time: nothing written, nothing counted, nothing signalled -- the plan has failed
    // behind the flags: what the senders stored before they raised them.  Hardware assumption of the default mode: the wave's
    // VMEM instructions issue in program order, so a load that stands behind the flag load is served behind it; the signal fence
    // keeps the COMPILER from hoisting the buffer loads (relaxed atomics, a control dependency only) above the wait -- the same
    // stale-face-with-a-ready-flag class as the round-3 defect, reintroduced silently by a scheduling change otherwise.
    __atomic_signal_fence(__ATOMIC_ACQUIRE);
    if (g.fenced) direct_acquire_fence();
#pragma unroll 1
    for (int l0 = 0; l0 < LG; l0 += CH) {
        if (k0 + l0 >= dK) break;
        auto level = [&](int l) { const int k = k0 + l0 + l; return k < dK ? k : dK - 1; };  // (levels past the last one: clamped, never stored)
        T c[CH][VEC], o[CH][VEC], w[CH], e[CH];
        U ghost[CH][VEC], ghost_w[CH], ghost_e[CH];
#pragma unroll
        for (int l = 0; l < CH; ++l) {
            T* const row = in.p + (int64_t)level(l) * in.sk + (int64_t)j * in.sj + i0;
            vload<T, VEC>(row, c[l]);
            vload<T, VEC>(in.p + (int64_t)level(l) * in.sk + (int64_t)far * in.sj + i0, o[l]);
            const U* buf = static_cast<const U*>(g.buffer[f]) + (int64_t)level(l) * g.ext[f] - g.lo[f];  // buf[i]: ghost value of column i
#pragma unroll
            for (int v = 0; v < VEC; ++v) ghost[l][v] = edge_load(buf + i0 + v);
            ghost_w[l] = corner_w ? edge_load(buf - 1) : (U)0;
            ghost_e[l] = corner_e ? edge_load(buf + dI) : (U)0;
            w[l] = lane_shift<T, true>(c[l][VEC - 1]);
            e[l] = lane_shift<T, false>(c[l][0]);
            // the W / E neighbour of the row's first / last point: the field's own ghost cell (a physical boundary, or what an
            // earlier phase put there) -- or, where a W / E face takes part in THIS exchange, its value from that face's buffer
            // (the column units write it into the field)
            if (lane == 0 && !need_w) w[l] = row[-1];
            if ((lane == 63 || i0 + VEC >= dI) && !(need_e && last)) e[l] = row[VEC];
            if (need_w && first) w[l] = __builtin_bit_cast(T, edge_load(static_cast<const U*>(g.buffer[0]) + (int64_t)level(l) * g.ext[0] + (j - g.lo[0])));
            if (need_e && last) e[l] = __builtin_bit_cast(T, edge_load(static_cast<const U*>(g.buffer[1]) + (int64_t)level(l) * g.ext[1] + (j - g.lo[1])));
        }
#pragma unroll
        for (int l = 0; l < CH; ++l) {
            const int k = k0 + l0 + l;
            if (!active || k >= dK) continue;
            T gv[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) gv[v] = __builtin_bit_cast(T, ghost[l][v]);
            T* const grow = in.p + (int64_t)k * in.sk + (int64_t)gj * in.sj;
            vstore<T, VEC, false>(grow + i0, gv);  // the exchange's contract: the field has its ghost cells
            if (corner_w) grow[-1] = __builtin_bit_cast(T, ghost_w[l]);
            if (corner_e) grow[dI] = __builtin_bit_cast(T, ghost_e[l]);
            T res[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T wv = (v == 0) ? w[l] : c[l][v - 1];
                const T ev = (v == VEC - 1) ? e[l] : c[l][v + 1];
                const T south = f == 2 ? gv[v] : o[l][v], north = f == 2 ? o[l][v] : gv[v];  // rows j - 1 and j + 1
                res[v] = lap5_expr<T, W, VARIANT>(c[l][v], wv, ev, south, north);
            }
            vstore<T, VEC, true>(out.p + (int64_t)k * out.sk + (int64_t)j * out.sj + i0, res);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every load of the buffers has returned
    return (1u << f) | (need_w ? 1u : 0u) | (need_e ? 2u : 0u);
}

// 64 rows of the W (f = 0) or E (f = 1) face of EDGE_LEVELS levels: a lane per row, CH levels at a time.  The lane computes the
// first (last) VEC columns of its row -- ONE whole 16-byte lane of the interior kernel, which therefore stores nothing there (its
// MASKED range is [VEC, dI - VEC)) -- and writes them with one 16-byte store: with one column here and the rest of that vector
// stored item by item by the interior, the 128 x 256 x 512 share paid three 8-byte partial-line stores per row and level
// (9 us of its 78).
template <typename T, typename W, int VARIANT, int VEC, int CH, int F>
__device__ __forceinline__ unsigned lap5_edge_col_unit(const View<T>& in, const View<T>& out, int dI, int dJ, int dK, const EdgeFaces& g,
                                                       const unsigned unit) {
    using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    constexpr int f = F;
    constexpr int LG = EDGE_LEVELS;
    const unsigned tiles_j = g.tiles[f];
    const int k0 = (int)(unit / tiles_j) * LG;
    const int lane = (int)(threadIdx.x & 63);
    // the face's rows [lo, lo + ext) get their ghost cells; the points of rows [rows_lo, rows_hi) are computed here
    int j = g.lo[f] + (int)(unit % tiles_j) * 64 + lane;
    const bool in_face = j < g.lo[f] + g.ext[f];
    if (!in_face) j = g.lo[f] + g.ext[f] - 1;
    const bool compute = in_face && j >= g.rows_lo && j < g.rows_hi;
    const int i0 = f == 0 ? 0 : dI - VEC, ig = f == 0 ? -1 : dI, ix = f == 0 ? VEC : dI - VEC - 1;  // the vector, the ghost column, the column beyond
    if (!edge_wait(g.wait_flag[f], g.wait_value[f], g.timeout_ticks, g.error)) return 0u;
    __atomic_signal_fence(__ATOMIC_ACQUIRE);  // (see lap5_edge_row_unit: the buffer loads stay behind the flag, in the compiler too)
    if (g.fenced) direct_acquire_fence();
#pragma unroll 1
    for (int l0 = 0; l0 < LG; l0 += CH) {
        if (k0 + l0 >= dK) break;
        auto level = [&](int l) { const int k = k0 + l0 + l; return k < dK ? k : dK - 1; };
        T c[CH][VEC], x[CH], s[CH][VEC], n[CH][VEC];  // the points, their neighbour beyond the vector, the rows below and above
        U ghost[CH];
#pragma unroll
        for (int l = 0; l < CH; ++l) {
            const T* const row = in.p + (int64_t)level(l) * in.sk + (int64_t)j * in.sj;
            vload<T, VEC>(row + i0, c[l]);
            x[l] = row[ix];
            ghost[l] = edge_load(static_cast<const U*>(g.buffer[f]) + (int64_t)level(l) * g.ext[f] + (j - g.lo[f]));
            const bool load_n = lane == 63 || !in_face || j + 1 >= g.lo[f] + g.ext[f];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                s[l][v] = lane_shift<T, true>(c[l][v]);   // row j - 1: the lane below
                n[l][v] = lane_shift<T, false>(c[l][v]);  // row j + 1
            }
            // (only for rows that are computed: a face that carries the ghost rows -1 / dJ has no row beyond them)
            if (lane == 0 && compute) vload<T, VEC>(row + i0 - in.sj, s[l]);
            if (load_n && compute) vload<T, VEC>(row + i0 + in.sj, n[l]);
        }
#pragma unroll
        for (int l = 0; l < CH; ++l) {
            const int k = k0 + l0 + l;
            if (!in_face || k >= dK) continue;
            const T gv = __builtin_bit_cast(T, ghost[l]);
            in.p[(int64_t)k * in.sk + (int64_t)j * in.sj + ig] = gv;  // the exchange's contract: the field has its ghost cells
            if (compute) {
                T res[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const T wv = v == 0 ? (f == 0 ? gv : x[l]) : c[l][v - 1];
                    const T ev = v == VEC - 1 ? (f == 0 ? x[l] : gv) : c[l][v + 1];
                    res[v] = lap5_expr<T, W, VARIANT>(c[l][v], wv, ev, s[l][v], n[l][v]);
                }
                vstore<T, VEC, true>(out.p + (int64_t)k * out.sk + (int64_t)j * out.sj + i0, res);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return 1u << f;
}

// Unit `unit` of the launch (a wave): which side it belongs to, and off it goes.
template <typename T, typename W, int VARIANT, int VEC, int CH>
__device__ __forceinline__ void lap5_edge_unit(const View<T>& in, const View<T>& out, int dI, int dJ, int dK, const EdgeFaces& g, unsigned unit) {
    unsigned mask = 0;  // the buffers this wave has read
    if (unit >= g.first[4]) mask = 0;
    else if (unit < g.first[1]) mask = lap5_edge_col_unit<T, W, VARIANT, VEC, CH, 0>(in, out, dI, dJ, dK, g, unit);
    else if (unit < g.first[2]) mask = lap5_edge_col_unit<T, W, VARIANT, VEC, CH, 1>(in, out, dI, dJ, dK, g, unit - g.first[1]);
    else if (unit < g.first[3]) mask = lap5_edge_row_unit<T, W, VARIANT, VEC, CH, 2>(in, out, dI, dJ, dK, g, unit - g.first[2]);
    else mask = lap5_edge_row_unit<T, W, VARIANT, VEC, CH, 3>(in, out, dI, dJ, dK, g, unit - g.first[3]);
    edge_block_done(g, mask);
}

// The messages of the phase that no unit reads (corner boxes): plain copies into the ghost cells (direct_block).
struct EdgeCopies {
    unsigned blocks;   // workgroups of the launch that copy: per_box for each of b.n boxes
    unsigned per_box;
    unsigned mask;     // bit m: box m of the batch is copied here (the others are the units' faces)
    BoxBatch b;
    DirectBatch d;
};

template <typename T, typename W, int VARIANT, int VEC, typename U>
__global__ void __launch_bounds__(256)
lap5_edge_kernel(View<T> in, View<T> out, int dI, int dJ, int dK, EdgeFaces g, EdgeCopies cp, U* field, int64_t si, int64_t sj, int64_t sk) {
    if (blockIdx.x < cp.blocks) {
        const unsigned m = blockIdx.x / cp.per_box;
        if (cp.mask >> m & 1u) direct_block<U, false>(field, si, sj, sk, cp.b, cp.d, (int)m, blockIdx.x % cp.per_box);
        return;
    }
    lap5_edge_unit<T, W, VARIANT, VEC, EDGE_LEVELS>(in, out, dI, dJ, dK, g, (blockIdx.x - cp.blocks) * 4 + (threadIdx.x >> 6));
}

// ONE launch per distributed apply (one-stream schedule, direct transport): the first workgroups push this rank's faces
// (direct_block<U, true>; padded to a multiple of 8 so that the interior's workgroups keep their XCD), then the interior's
// strips (lap5_strip_lane, the tiles and XCD-aware order of lap5_strip_kernel; MASKED: the first / last column of the view is
// not stored), and the copies and the edge units -- not last: after `split` of the interior's workgroups, so that the units'
// own latency (a flag, then dependent loads and stores: a few us, which as the very last workgroups of the launch
// they add to it in full) runs next to the rest of the interior.
// Their faces were pushed by the neighbours' FIRST workgroups and have long arrived by then.
template <typename T, typename W, int VARIANT, typename U, int TPB, bool MASKED>
__global__ void __launch_bounds__(256)
lap5_step_kernel(View<const T> in, View<T> out, int dI, int dJ_int, unsigned tiles_x, unsigned tiles_y, unsigned interior_tiles,
                 unsigned interior_blocks, unsigned push_pad, unsigned push_per_box, unsigned split, unsigned tail_pad, int c_lo, int c_hi,
                 U* field, int64_t si, int64_t sj, int64_t sk, BoxBatch pb, DirectBatch pd, View<T> in_dom, View<T> out_dom, int dJ, int dK,
                 EdgeFaces g, EdgeCopies cp) {
    constexpr int VEC = 16 / (int)sizeof(T), LJ = Lap5Tuning::LJ, LANES = 256 / TPB;
    if (blockIdx.x < push_pad) {
        const unsigned m = blockIdx.x / push_per_box;
        if (m < (unsigned)pb.n) direct_block<U, true>(field, si, sj, sk, pb, pd, (int)m, blockIdx.x % push_per_box);
        return;
    }
    // [0, split): interior | [split, split + tail_pad): copies and edge units | the rest of the interior (tail_pad is a multiple
    // of 8: the interior's workgroups keep their XCD)
    unsigned rel = blockIdx.x - push_pad;
    const bool is_tail = rel >= split && rel < split + tail_pad;
    if (!is_tail) {
        if (rel >= split) rel -= tail_pad;
        const unsigned w = xcd_remap_grouped<(unsigned)Lap5Tuning::XCDG>(rel, interior_blocks);
        const unsigned t = w * TPB + threadIdx.x / LANES;  // (whole waves: LANES is a multiple of 64)
        if (t >= interior_tiles) return;
        const unsigned bx = t % tiles_x, by = (t / tiles_x) % tiles_y, k = t / (tiles_x * tiles_y);
        const unsigned lane = threadIdx.x & 63;
        int i0 = (int)(bx * LANES + threadIdx.x % LANES) * VEC;
        const bool active = i0 < dI;
        if (!active) i0 = dI - VEC;
        lap5_strip_lane<T, W, VARIANT, VEC, LJ, 0, MASKED>(in, out, dJ_int, i0, active, lane == 0, (lane == 63) || (i0 + VEC >= dI),
                                                          (int)by * LJ, k, c_lo, c_hi);
        return;
    }
    const unsigned tail = rel - split;
    if (tail < cp.blocks) {
        const unsigned m = tail / cp.per_box;
        if (cp.mask >> m & 1u) direct_block<U, false>(field, si, sj, sk, cp.b, cp.d, (int)m, tail % cp.per_box);
        return;
    }
    lap5_edge_unit<T, W, VARIANT, VEC, EDGE_LEVELS>(in_dom, out_dom, dI, dJ, dK, g, (tail - cp.blocks) * 4 + (threadIdx.x >> 6));
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
// The faces and copies of `plan`'s (single) receiving phase for a halo-1 apply on `domain`; *ok = false when the plan, the shapes
// or the layout do not qualify (the caller keeps the unpack + ring launches).  `flags`: the direct transport's real flags, or
// (RCCL: the units run behind the send/recv kernel in stream order) words that are always satisfied.
template <typename T>
inline int lap5_edge_prepare(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* outf, int sides,
                             View<T>* in_v, View<T>* out_v, EdgeFaces* g, EdgeCopies* cp, int* phase_out, bool* ok) {
    using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    *ok = false;
    auto& dx = plan->direct;
    const bool direct = plan->transport == GT4MI_TRANSPORT_DIRECT;
    const int64_t di = domain[0], dj = domain[1], dk = domain[2];
    constexpr int VEC = 16 / (int)sizeof(T);
    static const int enabled = env_int("GT4MI_DIST_EDGE_UNITS", 1);
    if (!enabled || (direct && !dx.prepared) || plan->edge_words == nullptr) return GT4MI_OK;
    if (sides == 0 || di < 2 * VEC || dj < 2 || dk <= 0 || di % VEC != 0) return GT4MI_OK;
    int phase = -1;
    for (int p = 0; p < 2; ++p)
        if (!plan->recvs[p].empty() || !plan->sends[p].empty()) {
            if (phase >= 0) return GT4MI_OK;  // two rounds: the second one's faces depend on the first one's unpack
            phase = p;
        }
    if (phase < 0 || plan->recvs[phase].empty() || (int)plan->recvs[phase].size() > BoxBatch::MAX) return GT4MI_OK;
    const int h0[3] = {0, 0, 0}, h1[3] = {1, 1, 0};
    if (int rc = make_view<T>("inp", inp, domain, h1, h1, in_v)) return rc;
    if (int rc = make_view<T>("out", outf, domain, h0, h0, out_v)) return rc;
    if (!(in_v->si == 1 && out_v->si == 1 && vec_ok(View<const T>{in_v->p, 1, in_v->sj, in_v->sk}, VEC) && vec_ok(*out_v, VEC))) return GT4MI_OK;
    // (RCCL) flags that are always satisfied, a sink for the signals, counters of their own: plan->edge_words
    uint32_t* const zero = plan->edge_words, * const sink = plan->edge_words + 1;
    unsigned* const counters = direct ? dx.ring_counters : reinterpret_cast<unsigned*>(plan->edge_words + 4);
    for (int s = 0; s < 4; ++s) {
        g->have[s] = 0;
        g->buffer[s] = nullptr;
        g->lo[s] = g->ext[s] = 0;
        g->wait_flag[s] = g->consumed_flag[s] = zero;
        g->wait_value[s] = g->consumed_add[s] = 0;
        g->counter[s] = counters + s;
        g->readers[s] = 1;
        g->tiles[s] = 1;
    }
    if (int rc = direct_batches_recv<U>(plan, inp, phase, cp->b, cp->d)) return rc;
    cp->mask = 0;
    cp->per_box = 0;
    const unsigned groups = (unsigned)cdiv(dk, (int64_t)EDGE_LEVELS);
    for (size_t m = 0; m < plan->recvs[phase].size(); ++m) {
        const auto& msg = plan->recvs[phase][m];
        const int64_t jlo = msg.lo[1] - inp->origin[1], ilo = msg.lo[0] - inp->origin[0];
        const bool whole_k = msg.lo[2] == inp->origin[2] && msg.ext[2] == dk;
        int s = -1;
        if (whole_k && msg.ext[1] == 1 && (jlo == -1 || jlo == dj) && ilo <= 0 && ilo >= -1 && ilo + msg.ext[0] >= di && ilo + msg.ext[0] <= di + 1)
            s = jlo == -1 ? 2 : 3;
        else if (whole_k && msg.ext[0] == 1 && (ilo == -1 || ilo == di) && jlo <= 0 && jlo >= -1 && jlo + msg.ext[1] >= dj &&
                 jlo + msg.ext[1] <= dj + 1)  // (a face may carry the ghost rows of a side without a neighbour: rows -1 and / or dJ)
            s = ilo == -1 ? 0 : 1;
        if (s >= 0 && !g->have[s] && (sides >> s & 1)) {
            g->have[s] = 1;
            g->buffer[s] = msg.buffer;
            g->lo[s] = (int)(s < 2 ? jlo : ilo);
            g->ext[s] = (int)(s < 2 ? msg.ext[1] : msg.ext[0]);
            if (direct) {
                if (dx.signal_consumed[phase][m] == nullptr) return GT4MI_OK;
                const unsigned nb = direct_blocks(msg.bytes);
                g->wait_flag[s] = dx.flags + direct_index(plan, false, phase, (int)m);
                g->wait_value[s] = dx.step * nb;
                g->consumed_flag[s] = dx.signal_consumed[phase][m];
                g->consumed_add[s] = nb;
            } else {
                g->consumed_flag[s] = sink;
            }
        } else {  // a corner box (or anything else): copied as it is
            cp->mask |= 1u << m;
            cp->per_box = cp->d.blocks[m] > cp->per_box ? cp->d.blocks[m] : cp->per_box;
        }
    }
    for (int s = 0; s < 4; ++s)
        if ((sides >> s & 1) && !g->have[s]) return GT4MI_OK;  // a side with a neighbour but no face of the expected shape: not ours
    cp->blocks = cp->mask ? cp->per_box * (unsigned)cp->b.n : 0u;
    g->rows_lo = g->have[2] ? 1 : 0;
    g->rows_hi = (int)dj - (g->have[3] ? 1 : 0);
    unsigned first = 0;
    for (int s = 0; s < 4; ++s) {
        g->first[s] = first;
        if (!g->have[s]) continue;
        g->tiles[s] = (unsigned)(s < 2 ? cdiv((int64_t)g->ext[s], (int64_t)64) : cdiv(di, (int64_t)64 * VEC));
        const unsigned units = g->tiles[s] * groups;
        g->readers[s] = units;
        first += units;
    }
    g->first[4] = first;
    for (int s = 2; s < 4; ++s)  // the row units at either end of the S / N row read the W / E buffers too
        if (g->have[s]) {
            if (g->have[0]) g->readers[0] += groups;
            if (g->have[1]) g->readers[1] += groups;
        }
    g->error = direct ? dx.error : plan->edge_words + 2;
    g->timeout_ticks = direct_timeout_ticks(plan);
    g->fenced = direct ? dx.fenced : 0;
    *phase_out = phase;
    *ok = true;
    return GT4MI_OK;
}

// The copies and the edge units as a launch of their own on `stream` (behind the send/recv kernel of an RCCL plan, or behind
// the pushes of a direct one).  *done = false: nothing launched.
template <typename T, typename W>
inline int lap5_edge_run(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* outf, int variant,
                         int sides, hipStream_t stream, bool* done) {
    using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    *done = false;
    View<T> in_v, out_v;
    EdgeFaces g;
    EdgeCopies cp;
    int phase = 0;
    bool ok = false;
    if (int rc = lap5_edge_prepare<T>(plan, domain, inp, outf, sides, &in_v, &out_v, &g, &cp, &phase, &ok)) return rc;
    if (!ok) return GT4MI_OK;
    constexpr int VEC = 16 / (int)sizeof(T);
    const unsigned blocks = cp.blocks + (unsigned)cdiv((int64_t)g.first[4], (int64_t)4);
#define GT4MI_LAP5_EDGE(V)                                                                                                       \
    hipLaunchKernelGGL((lap5_edge_kernel<T, W, V, VEC, U>), dim3(blocks), dim3(256), 0, stream, in_v, out_v, (int)domain[0],      \
                       (int)domain[1], (int)domain[2], g, cp, static_cast<U*>(inp->data), inp->stride[0] / (int64_t)sizeof(U),   \
                       inp->stride[1] / (int64_t)sizeof(U), inp->stride[2] / (int64_t)sizeof(U))
    switch (variant) {
        case GT4MI_LAP_NOTEBOOK: GT4MI_LAP5_EDGE(GT4MI_LAP_NOTEBOOK); break;
        case GT4MI_LAP_DOCS: GT4MI_LAP5_EDGE(GT4MI_LAP_DOCS); break;
        case GT4MI_LAP_SUITE: GT4MI_LAP5_EDGE(GT4MI_LAP_SUITE); break;
        case GT4MI_LAP_AVG: GT4MI_LAP5_EDGE(GT4MI_LAP_AVG); break;
        default: return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5: unknown variant %d", variant);
    }
#undef GT4MI_LAP5_EDGE
    GT4MI_HIP_CHECK(hipGetLastError());
    *done = true;
    return GT4MI_OK;
}

// Push, interior and edge of one apply in ONE launch (direct transport, one-stream schedule).  The caller has advanced the plan's
// exchange counter (plan->direct.step) already; *done = false: nothing launched.
template <typename T, typename W>
inline int lap5_step_run(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* outf, int variant,
                         int sides, hipStream_t stream, bool* done) {
    using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    *done = false;
    if (plan->transport != GT4MI_TRANSPORT_DIRECT) return GT4MI_OK;
    View<T> in_v, out_v;
    EdgeFaces g;
    EdgeCopies cp;
    int phase = 0;
    bool ok = false;
    if (int rc = lap5_edge_prepare<T>(plan, domain, inp, outf, sides, &in_v, &out_v, &g, &cp, &phase, &ok)) return rc;
    if (!ok) return GT4MI_OK;
    constexpr int VMAX = 16 / (int)sizeof(T), LJ = Lap5Tuning::LJ;
    const int64_t di = domain[0], dj = domain[1], dk = domain[2];
    const int h1[3] = {1, 1, 0}, h0[3] = {0, 0, 0};
    if (views_overlap(in_v, h1, h1, out_v, h0, h0, domain))
        return fail(GT4MI_ERR_UNSUPPORTED, "lap5: 'inp' and 'out' overlap in memory (see gt4mi_lap5_*)");
    // the interior: every column of the rows that read no S / N ghost row; the first / last column is not stored where a W / E face
    // is on its way
    const int64_t lo_j = g.have[2] ? 1 : 0, rows = dj - lo_j - (g.have[3] ? 1 : 0);
    const View<const T> in_i{in_v.p + lo_j * in_v.sj, 1, in_v.sj, in_v.sk};
    const View<T> out_i{out_v.p + lo_j * out_v.sj, 1, out_v.sj, out_v.sk};
    const bool masked = g.have[0] || g.have[1];
    const int c_lo = g.have[0] ? VMAX : 0, c_hi = (int)di - (g.have[1] ? VMAX : 0);  // (whole lanes: lap5_edge_col_unit)
    const int64_t lanes_per_row = di / VMAX;
    const int tpb = lanes_per_row <= 64 ? 4 : (lanes_per_row <= 128 ? 2 : 1);  // tiles of 64 / 128 / 256 lanes, as lap5_launch_variant
    const unsigned tx = (unsigned)cdiv(di, (int64_t)(256 / tpb) * VMAX), ty = (unsigned)cdiv(rows > 0 ? rows : 0, (int64_t)LJ);
    const int64_t tiles = (int64_t)tx * ty * dk, interior = cdiv(tiles, (int64_t)tpb);
    BoxBatch pb;
    DirectBatch pd;
    int64_t per_box = 0;
    if (int rc = direct_batches<U, true>(plan, inp, phase, pb, pd, per_box)) return rc;
    const int64_t pad = cdiv(per_box * pb.n, (int64_t)8) * 8;
    const int64_t tail = cdiv((int64_t)cp.blocks + cdiv((int64_t)g.first[4], (int64_t)4), (int64_t)8) * 8;
    if (tiles > INT32_MAX || interior + pad + tail > INT32_MAX) return GT4MI_OK;
    // the units start after this share of the interior's workgroups (GT4MI_DIST_EDGE_AFTER_PERCENT; 100: as the last workgroups)
    static const int after = env_int("GT4MI_DIST_EDGE_AFTER_PERCENT", 85);
    const int64_t split = (interior * (after < 0 ? 0 : (after > 100 ? 100 : after)) / 100) / 8 * 8;
    const dim3 grid((unsigned)(pad + interior + tail));
#define GT4MI_LAP5_STEP_T(V, TPB, M)                                                                                              \
    hipLaunchKernelGGL((lap5_step_kernel<T, W, V, U, TPB, M>), grid, dim3(256), launch_dynamic_lds(), stream, in_i, out_i, (int)di, \
                       (int)rows, tx, ty, (unsigned)tiles, (unsigned)interior, (unsigned)pad, (unsigned)(per_box > 0 ? per_box : 1), \
                       (unsigned)split, (unsigned)tail, c_lo, c_hi, static_cast<U*>(inp->data), inp->stride[0] / (int64_t)sizeof(U),                               \
                       inp->stride[1] / (int64_t)sizeof(U), inp->stride[2] / (int64_t)sizeof(U), pb, pd, in_v, out_v, (int)dj,    \
                       (int)dk, g, cp)
#define GT4MI_LAP5_STEP(V)                                                    \
    do {                                                                      \
        if (masked) {                                                         \
            if (tpb == 1) GT4MI_LAP5_STEP_T(V, 1, true);                      \
            else if (tpb == 2) GT4MI_LAP5_STEP_T(V, 2, true);                 \
            else GT4MI_LAP5_STEP_T(V, 4, true);                               \
        } else {                                                              \
            if (tpb == 1) GT4MI_LAP5_STEP_T(V, 1, false);                     \
            else if (tpb == 2) GT4MI_LAP5_STEP_T(V, 2, false);                \
            else GT4MI_LAP5_STEP_T(V, 4, false);                              \
        }                                                                     \
    } while (0)
    switch (variant) {
        case GT4MI_LAP_NOTEBOOK: GT4MI_LAP5_STEP(GT4MI_LAP_NOTEBOOK); break;
        case GT4MI_LAP_DOCS: GT4MI_LAP5_STEP(GT4MI_LAP_DOCS); break;
        case GT4MI_LAP_SUITE: GT4MI_LAP5_STEP(GT4MI_LAP_SUITE); break;
        case GT4MI_LAP_AVG: GT4MI_LAP5_STEP(GT4MI_LAP_AVG); break;
        default: return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5: unknown variant %d", variant);
    }
#undef GT4MI_LAP5_STEP
#undef GT4MI_LAP5_STEP_T
    GT4MI_HIP_CHECK(hipGetLastError());
    *done = true;
    return GT4MI_OK;
}

}  // namespace gt4mi
