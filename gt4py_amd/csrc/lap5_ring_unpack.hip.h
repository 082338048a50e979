// The unpack of the direct transport and the ring of a distributed 5-point step in ONE launch -- for local domains cut along J
// only (S / N neighbours, no W / E ones: the 1 x N process grids).
//
// Behind the interior kernel of the one-stream ("inline") schedule two kernels ran in a row on an otherwise idle device: the
// unpack (receive buffers -> ghost rows, 8 us) and the ring (the rows that read them, 5 us).  The ring reads what the unpack
// writes, so the two cannot simply share a launch -- unless every ring tile fetches ITS part of the ghost row from the receive
// buffer itself: a wave owns 64 x VEC columns of one level of the first or last row of the domain; it loads the ghost values for
// its columns past the caches (they were written by another agent) AFTER one lane of the workgroup has seen the arrival flag
// (message passing: payload loads behind the flag load, direct.hip.h "ordering"; the loads of the field's own rows, which do not
// depend on the flag, are in flight meanwhile), writes them into the field's ghost cells (the exchange's contract: `inp` has its ghost cells afterwards; the wave at either
// end of the row also the corner columns a face may carry), and computes its points from them -- the same expression
// (lap5_expr) on the same values as the two-launch form, bit for bit.  The last workgroup of a face tells the sender that
// its buffer is free again (as many counts as the plain unpack kernel would have added: the flags count 16 KB blocks).
#pragma once

#include "direct.hip.h"
#include "lap5.hip.h"

#pragma clang fp contract(off)

namespace gt4mi {

struct RowGhosts {
    int n;                        // faces: 1 or 2 (S, N)
    int row[2], ghost_row[2];     // the domain row that reads the face (0 / dJ - 1) and the ghost row it fills (-1 / dJ)
    int ilo[2], ext_i[2];         // the face's columns [ilo, ilo + ext_i), relative to the domain's first column
    const void* buffer[2];        // dense: one row of ext_i items per level
    uint32_t* wait_flag[2];
    uint32_t wait_value[2];
    uint32_t* consumed_flag[2];   // at the sender
    uint32_t consumed_add[2];
    unsigned* counter[2];         // device: workgroups of this face that have read their part
    unsigned blocks[2];
    uint32_t* error;              // host memory mapped into the device
    long long timeout_ticks;
};

constexpr int RING_UNPACK_LEVELS = 8;  // K levels per wave: all their loads in flight at once, and an eighth of the workgroups
// (each one polls a flag and counts itself in on ONE word: with a workgroup per four levels the 1 024 same-address atomics of a
// 512-level share alone took longer than the two kernels this launch replaces)

template <typename T, typename W, int VARIANT, int VEC>
__global__ void __launch_bounds__(256)
lap5_ring_unpack_kernel(View<T> in, View<T> out, int dI, int dK, unsigned tiles_x, RowGhosts g) {
    using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    constexpr int LG = RING_UNPACK_LEVELS;
    const int f = blockIdx.y;
    if (blockIdx.x >= g.blocks[f]) return;
    __shared__ int ready;
    const unsigned groups = (unsigned)((dK + LG - 1) / LG);
    const unsigned unit = blockIdx.x * 4 + (threadIdx.x >> 6), units = tiles_x * groups;
    const bool live = unit < units;  // (idle waves of the last workgroup still take part in the barriers)
    const int k0 = live ? (int)(unit / tiles_x) * LG : 0;
    const unsigned tx = live ? unit % tiles_x : 0;
    const int lane = (int)(threadIdx.x & 63);
    int i0 = (int)(tx * 64 + lane) * VEC;
    const bool active = live && i0 < dI;
    if (i0 >= dI) i0 = dI - VEC;
    const int j = g.row[f], gj = g.ghost_row[f], far = (gj < j) ? j + 1 : j - 1;
    const bool first = active && i0 == 0, last = active && i0 + VEC == dI;
    const bool corner_w = first && g.ilo[f] < 0, corner_e = last && g.ilo[f] + g.ext_i[f] > dI;
    U ghost[LG][VEC], ghost_w[LG], ghost_e[LG];
    auto level = [&](int l) { const int k = k0 + l; return k < dK ? k : dK - 1; };  // (levels past the last one: clamped, never stored)
    auto load_ghosts = [&]() {
#pragma unroll
        for (int l = 0; l < LG; ++l) {
            const U* buf = static_cast<const U*>(g.buffer[f]) + (int64_t)level(l) * g.ext_i[f] - g.ilo[f];  // buf[i]: ghost value of column i
#pragma unroll
            for (int v = 0; v < VEC; ++v) ghost[l][v] = __hip_atomic_load(buf + i0 + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ghost_w[l] = corner_w ? __hip_atomic_load(buf - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : (U)0;
            ghost_e[l] = corner_e ? __hip_atomic_load(buf + dI, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : (U)0;
        }
    };
#ifdef GT4MI_DIRECT_ROUND3_LOAD_ORDER
    load_ghosts();  // (evidence build only, see direct_block: the ghost loads in front of the flag load)
#endif
    T c[LG][VEC], o[LG][VEC], w[LG], e[LG];  // the field's own rows: in flight while lane 0 waits for the face
#pragma unroll
    for (int l = 0; l < LG; ++l) {
        T* const row = in.p + (int64_t)level(l) * in.sk + (int64_t)j * in.sj + i0;
        vload<T, VEC>(row, c[l]);
        vload<T, VEC>(in.p + (int64_t)level(l) * in.sk + (int64_t)far * in.sj + i0, o[l]);
        w[l] = lane_shift<T, true>(c[l][VEC - 1]);
        e[l] = lane_shift<T, false>(c[l][0]);
        if (lane == 0) w[l] = row[-1];
        if (lane == 63 || i0 + VEC >= dI) e[l] = row[VEC];
    }
#ifdef GT4MI_DIRECT_ROUND3_LOAD_ORDER
    __shared__ int first_look;
    if (threadIdx.x == 0) {
        first_look = (int)(__hip_atomic_load(g.wait_flag[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - g.wait_value[f]) >= 0;
        ready = direct_wait(g.wait_flag[f], g.wait_value[f], g.timeout_ticks, g.error) ? 1 : 0;
    }
    __syncthreads();
    if (!ready) return;
    if (!first_look) load_ghosts();
#else
    if (threadIdx.x == 0) ready = direct_wait(g.wait_flag[f], g.wait_value[f], g.timeout_ticks, g.error) ? 1 : 0;
    __syncthreads();
    if (!ready) return;  // out of time: no ghost cell written, no point computed, nothing signalled -- the plan has failed
    load_ghosts();       // behind the flag: what the sender stored before it raised it
#endif
#pragma unroll
    for (int l = 0; l < LG; ++l) {
        const int k = k0 + l;
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
            const T south = (gj < j) ? gv[v] : o[l][v], north = (gj < j) ? o[l][v] : gv[v];  // rows j - 1 and j + 1
            res[v] = lap5_expr<T, W, VARIANT>(c[l][v], wv, ev, south, north);
        }
        vstore<T, VEC, true>(out.p + (int64_t)k * out.sk + (int64_t)j * out.sj + i0, res);
    }
    __syncthreads();  // every wave of the workgroup has its ghost values: this part of the buffer has been read
    if (threadIdx.x == 0) {
        const unsigned done = atomicInc(g.counter[f], g.blocks[f] - 1);  // wraps to 0 with the last workgroup
        if (done == g.blocks[f] - 1)
            __hip_atomic_fetch_add(g.consumed_flag[f], g.consumed_add[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The rest of the exchange of `inp` (whose faces were pushed already) and the ring, one launch.  *done = false, nothing launched,
// when the plan / the shapes do not qualify (the caller then unpacks and launches the ring as usual).
template <typename T, typename W>
inline int lap5_ring_unpack_run(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* outf,
                                int variant, int sides, hipStream_t stream, bool* done) {
    *done = false;
    auto& dx = plan->direct;
    const int64_t di = domain[0], dj = domain[1], dk = domain[2];
    constexpr int VEC = 16 / (int)sizeof(T);
    if (plan->transport != GT4MI_TRANSPORT_DIRECT || !dx.prepared || dx.ring_counters == nullptr) return GT4MI_OK;
    if ((sides & 3) || !(sides & 12) || dj < 2 || di <= 0 || dk <= 0 || di % VEC != 0) return GT4MI_OK;
    int phase = -1;
    for (int p = 0; p < 2; ++p)
        if (!plan->recvs[p].empty() || !plan->sends[p].empty()) {
            if (phase >= 0) return GT4MI_OK;  // two rounds: the second one's faces depend on the first one's unpack
            phase = p;
        }
    if (phase < 0 || plan->recvs[phase].empty() || plan->recvs[phase].size() > 2) return GT4MI_OK;
    const int h0[3] = {0, 0, 0}, h1[3] = {1, 1, 0};
    View<T> in_v, out_v;
    if (int rc = make_view<T>("inp", inp, domain, h1, h1, &in_v)) return rc;
    if (int rc = make_view<T>("out", outf, domain, h0, h0, &out_v)) return rc;
    if (!(in_v.si == 1 && out_v.si == 1 && vec_ok(View<const T>{in_v.p, 1, in_v.sj, in_v.sk}, VEC) && vec_ok(out_v, VEC))) return GT4MI_OK;
    RowGhosts g;
    g.n = 0;
    const unsigned tiles_x = (unsigned)cdiv(di, (int64_t)64 * VEC);
    const unsigned blocks = (unsigned)cdiv((int64_t)tiles_x * cdiv(dk, (int64_t)RING_UNPACK_LEVELS), (int64_t)4);
    for (size_t m = 0; m < plan->recvs[phase].size(); ++m) {
        const auto& msg = plan->recvs[phase][m];
        const int64_t jrow = msg.lo[1] - inp->origin[1], ilo = msg.lo[0] - inp->origin[0];
        const bool south = jrow == -1, north = jrow == dj;
        if (!(south || north) || msg.ext[1] != 1 || msg.lo[2] != inp->origin[2] || msg.ext[2] != dk) return GT4MI_OK;
        if (ilo > 0 || ilo < -1 || ilo + msg.ext[0] < di || ilo + msg.ext[0] > di + 1) return GT4MI_OK;
        if ((south && !(sides & 4)) || (north && !(sides & 8))) return GT4MI_OK;
        if (dx.signal_consumed[phase][m] == nullptr) return GT4MI_OK;
        const int f = g.n++;
        g.row[f] = south ? 0 : (int)dj - 1;
        g.ghost_row[f] = south ? -1 : (int)dj;
        g.ilo[f] = (int)ilo;
        g.ext_i[f] = (int)msg.ext[0];
        g.buffer[f] = msg.buffer;
        const unsigned nb = direct_blocks(msg.bytes);
        g.wait_flag[f] = dx.flags + direct_index(plan, false, phase, (int)m);
        g.wait_value[f] = dx.step * nb;
        g.consumed_flag[f] = dx.signal_consumed[phase][m];
        g.consumed_add[f] = nb;
        g.counter[f] = dx.ring_counters + f;
        g.blocks[f] = blocks;
    }
    if (g.n != (int)((sides & 4) != 0) + (int)((sides & 8) != 0)) return GT4MI_OK;  // (a side with a neighbour but no face: not ours)
    if (g.n == 2 && g.row[0] == g.row[1]) return GT4MI_OK;
    for (int f = g.n; f < 2; ++f) g.blocks[f] = 0;
    g.error = dx.error;
    g.timeout_ticks = direct_timeout_ticks(plan);
#define GT4MI_LAP5_RING_UNPACK(V)                                                                                                  \
    hipLaunchKernelGGL((lap5_ring_unpack_kernel<T, W, V, VEC>), dim3(blocks, (unsigned)g.n), dim3(256), 0, stream, in_v, out_v, (int)di, \
                       (int)dk, tiles_x, g)
    switch (variant) {
        case GT4MI_LAP_NOTEBOOK: GT4MI_LAP5_RING_UNPACK(GT4MI_LAP_NOTEBOOK); break;
        case GT4MI_LAP_DOCS: GT4MI_LAP5_RING_UNPACK(GT4MI_LAP_DOCS); break;
        case GT4MI_LAP_SUITE: GT4MI_LAP5_RING_UNPACK(GT4MI_LAP_SUITE); break;
        case GT4MI_LAP_AVG: GT4MI_LAP5_RING_UNPACK(GT4MI_LAP_AVG); break;
        default: return fail(GT4MI_ERR_INVALID_ARGUMENT, "lap5: unknown variant %d", variant);
    }
#undef GT4MI_LAP5_RING_UNPACK
    GT4MI_HIP_CHECK(hipGetLastError());
    dx.first_pushed = false;  // this exchange is complete
    *done = true;
    return GT4MI_OK;
}

}  // namespace gt4mi
