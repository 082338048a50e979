// libgt4py_amd.so -- extern "C" entry points declared in include/gt4py_amd.h.
//
// Each stencil entry replaces the pybind11 `run_computation` the reference generates per stencil
// (/root/reference/src/gt4py/cartesian/backend/gtc_common.py:65-103): validate the fields against
// the domain, shift pointers by the origins (gtc_common.py:48 `sid::shift_sid_origin`), launch the
// hand-written gfx950 kernel on the caller's stream.
#include <chrono>
#include <cstring>

#include "comm.hip.h"
#include "common.hip.h"
#include "halo.hip.h"
#include "memprobe.hip.h"
#include "hdiff.hip.h"
#include "hdiff_ring.hip.h"
#include "lap5.hip.h"
#include "lap5_push.hip.h"
#include "lap5_ring.hip.h"
#include "lap5_edge.hip.h"
#include "rtc.hip.h"
#include "tridiag.hip.h"

namespace gt4mi {
// How a fused distributed step is laid out on the two streams, and how far the interior kernel is throttled while the
// exchange runs next to it: the plan's options (gt4mi_halo_plan_set_option), else the environment (experiments), else the
// entry point's default -- measured on the 1-GPU self-loop, profiles/r3_dist_*_timeline*.txt.
inline int plan_schedule(const gt4mi_halo_plan* plan, int fallback) {
    static const int env = env_int("GT4MI_DIST_SCHEDULE", -1);
    return plan->schedule >= 0 ? plan->schedule : (env >= 0 ? env : fallback);
}
inline int plan_edge_columns(const gt4mi_halo_plan* plan, int fallback) {
    static const int env = env_int("GT4MI_DIST_EDGE_COLUMNS", -1);
    return plan->edge_columns >= 0 ? plan->edge_columns : (env >= 0 ? env : fallback);
}
inline int plan_interior_wg_per_cu(const gt4mi_halo_plan* plan, int fallback) {
    static const int env = env_int("GT4MI_DIST_INTERIOR_WG_PER_CU", -1);
    return plan->interior_wg_per_cu >= 0 ? plan->interior_wg_per_cu : (env >= 0 ? env : fallback);
}
}  // namespace gt4mi

namespace {

inline double now_seconds() {
    using clock = std::chrono::steady_clock;
    return std::chrono::duration<double>(clock::now().time_since_epoch()).count();
}

// Brackets one entry point: host timestamps of the (asynchronous) call and, from a hipEvent pair on the launch
// stream, the time its kernels spent on the device.  Costs nothing when the caller passes no exec_info.
struct Timer {
    gt4mi_exec_info* info;
    hipStream_t stream;
    Timer(gt4mi_exec_info* i, void* s) : info(i), stream(static_cast<hipStream_t>(s)) {
        if (!info) return;
        info->run_cpp_start_time = now_seconds();
        info->run_hip_start_time = info->run_hip_end_time = 0.0;
        if (events_ready()) (void)hipEventRecord(events()[0], stream);
    }
    ~Timer() {
        if (!info) return;
        info->run_cpp_end_time = now_seconds();
        if (!events_ready() || hipEventRecord(events()[1], stream) != hipSuccess ||
            hipEventSynchronize(events()[1]) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        float ms = 0.0f;
        const double end = now_seconds();
        if (hipEventElapsedTime(&ms, events()[0], events()[1]) == hipSuccess) {
            info->run_hip_end_time = end;
            info->run_hip_start_time = end - (double)ms * 1e-3;
        }
    }
    static hipEvent_t* events() {
        static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
        return ev;
    }
    static bool events_ready() {
        hipEvent_t* ev = events();
        if (ev[0] == nullptr)
            if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) {
                ev[0] = ev[1] = nullptr;
                (void)hipGetLastError();
            }
        return ev[0] != nullptr;
    }
};

}  // namespace

namespace {
template <typename T>
int dist_hdiff(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
               const gt4mi_field* coeff, double coeff_scalar, int flags, int sides, void* main_stream) {
    if (plan == nullptr || in_field == nullptr || out_field == nullptr || domain == nullptr)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_hdiff: null argument");
    if (plan->elem_size != (int)sizeof(T))
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_hdiff: the plan moves %d-byte items, the fields hold %d-byte items",
                           plan->elem_size, (int)sizeof(T));
    hipStream_t ms = static_cast<hipStream_t>(main_stream);
    if (int rc = gt4mi::direct_failed(plan)) return rc;
    if (int rc = gt4mi::ensure_concurrent_stream(plan, ms)) return rc;
    const int64_t di = domain[0], dj = domain[1], dk = domain[2];
    constexpr int64_t H = 2;  // the stencil's reach
    // W / E: the ring takes a box EW >= 2 columns wide (whole cache lines, J-march strips; hdiff_ring.hip.h) off the interior
    // kernel -- even, so that both parts keep their 16-byte alignment; only where the interior keeps at least as much
    int64_t EW = gt4mi::plan_edge_columns(plan, 16);
    EW = EW < H ? H : EW - EW % 2;
    if (di < 4 * EW) EW = H;
    int64_t lo_i = (sides & 1) ? EW : 0, hi_i = (sides & 2) ? EW : 0, lo_j = (sides & 4) ? H : 0, hi_j = (sides & 8) ? H : 0;
    lo_i = lo_i < di ? lo_i : di;
    hi_i = hi_i < di - lo_i ? hi_i : di - lo_i;
    lo_j = lo_j < dj ? lo_j : dj;
    hi_j = hi_j < dj - lo_j ? hi_j : dj - lo_j;
    // refuse what the ring would refuse BEFORE anything is enqueued (bounds, aliases): an empty ring validates only
    const int none[4] = {0, 0, 0, 0};
    if (int rc = gt4mi::hdiff_ring_run<T>(domain, in_field, out_field, coeff, coeff_scalar, flags, none, ms)) return rc;
    const int widths[4] = {(int)lo_i, (int)hi_i, (int)lo_j, (int)hi_j};
    auto interior = [&](hipStream_t st) -> int {
        if (!(di - lo_i - hi_i > 0 && dj - lo_j - hi_j > 0 && dk > 0)) return GT4MI_OK;
        gt4mi_field a = *in_field, b = *out_field, c;
        a.origin[0] += lo_i; a.origin[1] += lo_j;
        b.origin[0] += lo_i; b.origin[1] += lo_j;
        if (coeff) {
            c = *coeff;
            c.origin[0] += lo_i; c.origin[1] += lo_j;
        }
        const int64_t sub[3] = {di - lo_i - hi_i, dj - lo_j - hi_j, dk};
        // 2 of 4 workgroups per CU: the send/recv kernel next to it takes 59 us instead of 190 (3 of 4: 77;
        // profiles/r3_dist_hdiff_timeline_by_schedule_and_throttle.txt, r3_dist_hdiff_edge_width_sweep.txt)
        gt4mi::ScopedLaunchLds throttle(gt4mi::lds_for_workgroups_per_cu(gt4mi::plan_interior_wg_per_cu(plan, 2)));
        return gt4mi::hdiff_run<T>(sub, &a, &b, coeff ? &c : nullptr, coeff_scalar, flags, st);
    };
    const int schedule = gt4mi::plan_schedule(plan, GT4MI_SCHEDULE_CHAIN);
    if (schedule == GT4MI_SCHEDULE_INLINE) {  // one stream, no event (see dist_lap5)
        if (int rc = gt4mi::halo_pack_first(plan, in_field, ms)) return rc;
        if (int rc = interior(ms)) return rc;
        if (int rc = gt4mi::halo_exchange_on(plan, in_field, ms, /*first_pack_done=*/true)) return rc;
        return gt4mi::hdiff_ring_run<T>(domain, in_field, out_field, coeff, coeff_scalar, flags, widths, ms);
    }
    if (schedule == GT4MI_SCHEDULE_SWAP || schedule == GT4MI_SCHEDULE_SWAP_PACKED) {
        // Schedules "swap" / "swap-packed" (see gt4mi_dist_lap5_f64): the chain pack -> send/recv -> unpack -> ring back to back
        // on the CALLER's stream, the interior kernel on the side stream -- forked off before the pack, or after it so that the
        // send/recv kernel starts ahead of the interior's ramp-up; the caller's stream joins the interior at the end.
        if (schedule == GT4MI_SCHEDULE_SWAP_PACKED) {
            if (int rc = gt4mi::halo_pack_first(plan, in_field, ms)) return rc;
            GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
            GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
            if (int rc = gt4mi::halo_exchange_on(plan, in_field, ms, /*first_pack_done=*/true)) return rc;
            if (int rc = interior(plan->stream)) return rc;
        } else {
            GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
            GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
            if (int rc = interior(plan->stream)) return rc;
            if (int rc = gt4mi::halo_exchange_on(plan, in_field, ms)) return rc;
        }
        GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
        if (int rc = gt4mi::hdiff_ring_run<T>(domain, in_field, out_field, coeff, coeff_scalar, flags, widths, ms)) return rc;
        if (!plan->defer_join) GT4MI_HIP_CHECK(hipStreamWaitEvent(ms, plan->done, 0));
        return GT4MI_OK;
    }
    if (schedule == GT4MI_SCHEDULE_CHAIN) {
        // Schedule "chain": the main stream carries NOTHING but the interior kernel; pack -> send/recv -> unpack -> ring run
        // in order on the side stream (the ring writes out_field's ring, the interior its interior).  No cross-stream wait
        // lies on the critical path: the join after the interior is already satisfied when the chain fits under it, and
        // back-to-back applies run their interiors back to back (profiles/r3_dist_hdiff_timeline_*.txt).
        GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
        GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
        if (int rc = interior(ms)) return rc;
        if (int rc = gt4mi::halo_exchange_on(plan, in_field, plan->stream)) return rc;
        if (int rc = gt4mi::hdiff_ring_run<T>(domain, in_field, out_field, coeff, coeff_scalar, flags, widths, plan->stream)) return rc;
        GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
        if (!plan->defer_join) GT4MI_HIP_CHECK(hipStreamWaitEvent(ms, plan->done, 0));
        return GT4MI_OK;
    }
    // Schedule "join": 1. pack the first faces on the main stream, ahead of the interior kernel (alone: ~5 us; next to it: 20+)
    if (int rc = gt4mi::halo_pack_first(plan, in_field, ms)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
    GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
    // 2. main stream: the interior, which reads no ghost cell
    if (int rc = interior(ms)) return rc;
    // 3. side stream: send / receive / unpack (and the second phase of a two-phase plan) next to the interior kernel
    if (int rc = gt4mi::halo_exchange_on(plan, in_field, plan->stream, /*first_pack_done=*/true)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
    // 4. main stream: join, then the ring that reads the ghost cells -- one launch for all four boxes
    GT4MI_HIP_CHECK(hipStreamWaitEvent(ms, plan->done, 0));
    return gt4mi::hdiff_ring_run<T>(domain, in_field, out_field, coeff, coeff_scalar, flags, widths, ms);
}
// One distributed apply of a 5-point stencil (gt4mi_dist_lap5_f64 / _f32): T the fields' type, W the type its literals have.
template <typename T, typename W>
int dist_lap5(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant,
              int sides, void* main_stream) {
    if (plan == nullptr || inp == nullptr || out == nullptr || domain == nullptr)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_lap5: null argument");
    if (plan->elem_size != (int)sizeof(T))
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_lap5: the plan moves %d-byte items, the fields hold %d-byte items",
                           plan->elem_size, (int)sizeof(T));
    hipStream_t ms = static_cast<hipStream_t>(main_stream);
    if (int rc = gt4mi::direct_failed(plan)) return rc;
    if (int rc = gt4mi::ensure_concurrent_stream(plan, ms)) return rc;
    const int64_t di = domain[0], dj = domain[1], dk = domain[2];
    // W / E: the ring takes a box EW columns wide off the interior kernel (whole cache lines; and the interior then starts
    // on a 16-byte boundary -- one column in, it ran on 8-byte lanes at 85 us instead of 51 for the 128 x 256 x 512 share)
    int64_t EW = gt4mi::plan_edge_columns(plan, 16);
    EW = EW < 1 ? 1 : (EW > 16 ? 16 : EW);
    if (EW > 1) EW -= EW % (int64_t)(16 / sizeof(T));  // whole 16-byte lanes: the interior kernel keeps its alignment
    if (EW < 1) EW = 1;
    if (di < 16 * EW) EW = di >= 64 ? (EW < 8 ? EW : 8) : 1;  // narrow local domains keep most of their columns in the interior
    // Round 4: where the plan and the layout allow it the unpack and the ring are ONE kernel of wave-sized units that read the
    // receive buffers themselves (lap5_edge.hip.h); the interior then keeps every column but the first / last one (masked
    // 16-byte lanes: lap5_launch_variant) instead of giving 8-16 columns to a ring of 64-byte pieces.
    bool edge_units = false;
    {
        gt4mi::View<T> vi, vo;
        gt4mi::EdgeFaces g;
        gt4mi::EdgeCopies cp;
        int ph = 0;
        if (int rc = gt4mi::lap5_edge_prepare<T>(plan, domain, inp, out, sides, &vi, &vo, &g, &cp, &ph, &edge_units)) return rc;
    }
    if (edge_units) EW = 16 / (int64_t)sizeof(T);  // one 16-byte lane: what a column unit computes (lap5_edge.hip.h)
    const int64_t lo_i = (sides & 1) ? EW : 0, hi_i = (sides & 2) ? EW : 0;
    const int64_t lo_j = (sides & 4) ? 1 : 0, hi_j = (sides & 8) ? 1 : 0;
    auto run = [&](int64_t si, int64_t sj, int64_t ei, int64_t ej, hipStream_t st) -> int {
        if (ei <= 0 || ej <= 0 || dk <= 0) return GT4MI_OK;
        gt4mi_field a = *inp, b = *out;
        a.origin[0] += si; a.origin[1] += sj;
        b.origin[0] += si; b.origin[1] += sj;
        const int64_t d[3] = {ei, ej, dk};
        return gt4mi::lap5_run<T, W>(d, &a, &b, variant, st);
    };
    const int outer[4] = {0, 0, 0, 0};
    const int inner[4] = {(int)(lo_i <= di ? lo_i : di), (int)(hi_i && di - hi_i >= lo_i ? hi_i : 0), (int)lo_j,
                          (int)(hi_j && dj - 1 >= lo_j ? 1 : 0)};
    auto interior = [&](hipStream_t st) -> int {
        gt4mi::ScopedLaunchLds throttle(gt4mi::lds_for_workgroups_per_cu(gt4mi::plan_interior_wg_per_cu(plan, 0)));
        return run(lo_i, lo_j, di - lo_i - hi_i, dj - lo_j - hi_j, st);
    };
    // what follows the pack(s) on stream `st`: the rest of the exchange and the points that read ghost cells
    auto exchange_and_ring = [&](hipStream_t st, bool first_pack_done) -> int {
        if (edge_units) {
            if (int rc = gt4mi::halo_exchange_on(plan, inp, st, first_pack_done, /*skip_last_unpack=*/true)) return rc;
            bool done = false;
            if (int rc = gt4mi::lap5_edge_run<T, W>(plan, domain, inp, out, variant, sides, st, &done)) return rc;
            if (done) return GT4MI_OK;
            plan->direct.broken = plan->transport == GT4MI_TRANSPORT_DIRECT ? "the edge units of a fused step could not be launched" : nullptr;
            return gt4mi::fail(GT4MI_ERR_HIP, "dist_lap5: the edge units qualified before the exchange and no longer do");
        }
        if (int rc = gt4mi::halo_exchange_on(plan, inp, st, first_pack_done)) return rc;
        return gt4mi::lap5_ring_run<T, W>(domain, inp, out, variant, outer, inner, st);
    };
    // default: the fastest on every share of 8 ranks measured (1 x 8, 2 x 4, 4 x 2) -- "swap" with RCCL,
    // "inline" when the pack kernel is the transfer (direct transport)
    const int schedule = gt4mi::plan_schedule(plan, plan->transport == GT4MI_TRANSPORT_DIRECT ? GT4MI_SCHEDULE_INLINE : GT4MI_SCHEDULE_SWAP);
    if (schedule == GT4MI_SCHEDULE_INLINE) {
        // ONE stream, no event: pack (with the direct transport: the faces are on their way when it ends), the interior kernel,
        // then whatever is left of the exchange (direct: the unpack, whose data arrived long ago) and the ring
        if (int rc = gt4mi::lap5_ring_run<T, W>(domain, inp, out, variant, outer, outer, ms)) return rc;  // validates only
        bool fused = false;
        const int p0 = gt4mi::first_phase(plan);
        if (edge_units && plan->transport == GT4MI_TRANSPORT_DIRECT && p0 < 2) {
            // ONE launch: push | interior | copies and edge units (lap5_step_kernel)
            bool done = false;
            ++plan->direct.step;  // (what halo_pack_first does on this transport; the launch reads it)
            const int rc = gt4mi::lap5_step_run<T, W>(plan, domain, inp, out, variant, sides, ms, &done);
            if (rc || !done) --plan->direct.step;
            if (rc) return rc;
            if (done) {
                plan->direct.first_pushed = false;  // this exchange is complete
                return GT4MI_OK;
            }
        }
        if (!edge_units && plan->transport == GT4MI_TRANSPORT_DIRECT && p0 < 2) {
            // ... and with the direct transport the push rides in the interior's launch (lap5_push.hip.h): 8-9 us off the step
            gt4mi_field a = *inp, b = *out;
            a.origin[0] += lo_i; a.origin[1] += lo_j;
            b.origin[0] += lo_i; b.origin[1] += lo_j;
            const int64_t sub[3] = {di - lo_i - hi_i, dj - lo_j - hi_j, dk};
            ++plan->direct.step;  // (what halo_pack_first does on this transport)
            if (int rc = gt4mi::lap5_interior_with_push<T, W>(plan, sub, &a, &b, variant, inp, p0, ms, &fused)) {
                --plan->direct.step;
                return rc;
            }
            if (fused) plan->direct.first_pushed = true;
            else --plan->direct.step;
        }
        if (!fused) {
            if (int rc = gt4mi::halo_pack_first(plan, inp, ms)) return rc;
            if (int rc = interior(ms)) return rc;
        }
        return exchange_and_ring(ms, /*first_pack_done=*/true);
    }
    if (schedule == GT4MI_SCHEDULE_SWAP || schedule == GT4MI_SCHEDULE_SWAP_PACKED) {
        // the CALLER's stream carries the chain pack -> send/recv -> unpack -> ring (no cross-stream wait inside it, and it
        // starts at once); the interior kernel runs beside it on the side stream; the caller joins the interior at the end
        if (int rc = gt4mi::lap5_ring_run<T, W>(domain, inp, out, variant, outer, outer, ms)) return rc;  // validates only
        if (schedule == GT4MI_SCHEDULE_SWAP_PACKED) {
            // ... and the interior kernel forks off AFTER the pack: the send/recv kernel gets a head start on the interior's
            // ramp-up and the pack of strided I faces (8-10 us next to the interior) runs alone
            if (int rc = gt4mi::halo_pack_first(plan, inp, ms)) return rc;
            GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
            GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
            if (int rc = exchange_and_ring(ms, /*first_pack_done=*/true)) return rc;
            if (int rc = interior(plan->stream)) return rc;
            GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
            plan->done_recorded = true;
        } else {
            GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
            GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
            if (int rc = interior(plan->stream)) return rc;
            GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
            plan->done_recorded = true;
            if (int rc = exchange_and_ring(ms, /*first_pack_done=*/false)) return rc;
        }
        return plan->defer_join ? GT4MI_OK : gt4mi_halo_exchange_end(plan, main_stream);
    }
    if (schedule == GT4MI_SCHEDULE_CHAIN) {
        // the main stream carries the interior kernel only; pack -> send/recv -> unpack -> ring in order on the side stream
        // (see dist_hdiff)
        if (int rc = gt4mi::lap5_ring_run<T, W>(domain, inp, out, variant, outer, outer, ms)) return rc;  // validates only
        GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
        GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
        if (int rc = interior(ms)) return rc;
        if (int rc = exchange_and_ring(plan->stream, /*first_pack_done=*/false)) return rc;
        GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
        return plan->defer_join ? GT4MI_OK : gt4mi_halo_exchange_end(plan, main_stream);
    }
    // 1. pack the first faces ON THE MAIN STREAM, ahead of the interior kernel: alone it takes ~5 us;
    //    launched next to the interior kernel's thousands of workgroups it took 22 us and delayed the
    //    whole exchange past the end of the interior kernel (profiles/r1_dist_step_timeline.txt).
    //    The side stream then only waits for this pack (and whatever preceded it).
    if (int rc = gt4mi::halo_pack_first(plan, inp, ms)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
    GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
    // 2. main stream: interior, independent of the ghost cells in flight
    if (int rc = interior(ms)) return rc;
    // 3. side stream: RCCL send/recv + unpack (+ second phase), concurrent with the interior kernel
    if (int rc = gt4mi::halo_exchange_on(plan, inp, plan->stream, /*first_pack_done=*/true, /*skip_last_unpack=*/edge_units)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
    plan->done_recorded = true;
    // 4. main stream: join, then the points that read ghost cells -- ONE launch (lap5_edge.hip.h: the unpack rides along;
    //    else lap5_ring.hip.h)
    if (int rc = gt4mi_halo_exchange_end(plan, main_stream)) return rc;
    if (edge_units) {
        bool done = false;
        if (int rc = gt4mi::lap5_edge_run<T, W>(plan, domain, inp, out, variant, sides, ms, &done)) return rc;
        if (done) return GT4MI_OK;
        return gt4mi::fail(GT4MI_ERR_HIP, "dist_lap5: the edge units qualified before the exchange and no longer do");
    }
    return gt4mi::lap5_ring_run<T, W>(domain, inp, out, variant, outer, inner, ms);
}

}  // namespace

extern "C" {

int gt4mi_abi_version(void) { return GT4MI_ABI_VERSION; }

const char* gt4mi_last_error(void) { return gt4mi::error_buffer(); }

int gt4mi_device_info(char* buf, size_t buflen) {
    if (buf == nullptr || buflen == 0) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "device_info: null buffer");
    int dev = 0;
    GT4MI_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    GT4MI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "device=%d name=%s arch=%s cus=%d clock_mhz=%d mem_gib=%.1f", dev, prop.name,
             prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000,
             (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return GT4MI_OK;
}

int gt4mi_stream_sync(void* stream) {
    GT4MI_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return GT4MI_OK;
}

int gt4mi_lap5_f64(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant,
                   int flags, void* stream, gt4mi_exec_info* info) {
    (void)flags;
    Timer t(info, stream);
    return gt4mi::lap5_run<double, double>(domain, inp, out, variant, static_cast<hipStream_t>(stream));
}

int gt4mi_lap5_f32(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant,
                   int flags, void* stream, gt4mi_exec_info* info) {
    Timer t(info, stream);
    if (flags & GT4MI_LAP_LITERAL_F32)
        return gt4mi::lap5_run<float, float>(domain, inp, out, variant, static_cast<hipStream_t>(stream));
    return gt4mi::lap5_run<float, double>(domain, inp, out, variant, static_cast<hipStream_t>(stream));
}

int gt4mi_hdiff_f64(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                    const gt4mi_field* coeff, double coeff_scalar, int flags, void* stream,
                    gt4mi_exec_info* info) {
    Timer t(info, stream);
    return gt4mi::hdiff_run<double>(domain, in_field, out_field, coeff, coeff_scalar, flags,
                                    static_cast<hipStream_t>(stream));
}

int gt4mi_hdiff_f32(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                    const gt4mi_field* coeff, double coeff_scalar, int flags, void* stream,
                    gt4mi_exec_info* info) {
    Timer t(info, stream);
    return gt4mi::hdiff_run<float>(domain, in_field, out_field, coeff, coeff_scalar, flags,
                                   static_cast<hipStream_t>(stream));
}

int gt4mi_hdiff_ring_f64(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                         const gt4mi_field* coeff, double coeff_scalar, int flags, const int widths[4], void* stream,
                         gt4mi_exec_info* info) {
    Timer t(info, stream);
    return gt4mi::hdiff_ring_run<double>(domain, in_field, out_field, coeff, coeff_scalar, flags, widths,
                                         static_cast<hipStream_t>(stream));
}

int gt4mi_hdiff_ring_f32(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                         const gt4mi_field* coeff, double coeff_scalar, int flags, const int widths[4], void* stream,
                         gt4mi_exec_info* info) {
    Timer t(info, stream);
    return gt4mi::hdiff_ring_run<float>(domain, in_field, out_field, coeff, coeff_scalar, flags, widths,
                                        static_cast<hipStream_t>(stream));
}

int gt4mi_lap5_ring_f64(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant, int flags,
                        const int outer[4], const int inner[4], void* stream, gt4mi_exec_info* info) {
    (void)flags;
    Timer t(info, stream);
    return gt4mi::lap5_ring_run<double, double>(domain, inp, out, variant, outer, inner, static_cast<hipStream_t>(stream));
}

int gt4mi_lap5_ring_f32(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant, int flags,
                        const int outer[4], const int inner[4], void* stream, gt4mi_exec_info* info) {
    Timer t(info, stream);
    if (flags & GT4MI_LAP_LITERAL_F32)
        return gt4mi::lap5_ring_run<float, float>(domain, inp, out, variant, outer, inner, static_cast<hipStream_t>(stream));
    return gt4mi::lap5_ring_run<float, double>(domain, inp, out, variant, outer, inner, static_cast<hipStream_t>(stream));
}

int gt4mi_tridiag_f64(const int64_t domain[3], const gt4mi_field* inf, const gt4mi_field* diag,
                      const gt4mi_field* sup, const gt4mi_field* rhs, const gt4mi_field* out, void* stream,
                      gt4mi_exec_info* info) {
    Timer t(info, stream);
    return gt4mi::tridiag_run<double>(domain, inf, diag, sup, rhs, out, static_cast<hipStream_t>(stream));
}

int gt4mi_tridiag_f32(const int64_t domain[3], const gt4mi_field* inf, const gt4mi_field* diag,
                      const gt4mi_field* sup, const gt4mi_field* rhs, const gt4mi_field* out, void* stream,
                      gt4mi_exec_info* info) {
    Timer t(info, stream);
    return gt4mi::tridiag_run<float>(domain, inf, diag, sup, rhs, out, static_cast<hipStream_t>(stream));
}

int gt4mi_halo_pack(const gt4mi_field* field, const int64_t lo[3], const int64_t extent[3], void* buffer,
                    int elem_size, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (elem_size == 8) return gt4mi::halo_copy<uint64_t, true>(field, lo, extent, buffer, s);
    if (elem_size == 4) return gt4mi::halo_copy<uint32_t, true>(field, lo, extent, buffer, s);
    return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "halo_pack: element size %d", elem_size);
}

int gt4mi_halo_unpack(const gt4mi_field* field, const int64_t lo[3], const int64_t extent[3],
                      const void* buffer, int elem_size, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* b = const_cast<void*>(buffer);
    if (elem_size == 8) return gt4mi::halo_copy<uint64_t, false>(field, lo, extent, b, s);
    if (elem_size == 4) return gt4mi::halo_copy<uint32_t, false>(field, lo, extent, b, s);
    return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "halo_unpack: element size %d", elem_size);
}

// ---- multi-GPU ----------------------------------------------------------------------------------
int gt4mi_comm_unique_id(void* id128) {
    if (id128 == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "comm_unique_id: null buffer");
    if (!gt4mi::rccl().ok) return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "librccl could not be loaded");
    GT4MI_RCCL_CHECK(gt4mi::rccl().GetUniqueId(static_cast<gt4mi::RcclUniqueId*>(id128)));
    return GT4MI_OK;
}

int gt4mi_comm_create(const void* id128, int nranks, int rank, gt4mi_comm** comm) {
    if (id128 == nullptr || comm == nullptr || nranks < 1 || rank < 0 || rank >= nranks)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "comm_create: invalid argument");
    if (!gt4mi::rccl().ok) return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "librccl could not be loaded");
    gt4mi::RcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    gt4mi_comm* c = new gt4mi_comm;
    c->nranks = nranks;
    c->rank = rank;
    int r = gt4mi::rccl().CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) {
        delete c;
        return gt4mi::fail(GT4MI_ERR_HIP, "ncclCommInitRank failed: %s",
                           gt4mi::rccl().GetErrorString ? gt4mi::rccl().GetErrorString(r) : "rccl error");
    }
    *comm = c;
    return GT4MI_OK;
}

int gt4mi_comm_create_local(int nranks, int rank, gt4mi_comm** comm) {
    if (comm == nullptr || nranks < 1 || rank < 0 || rank >= nranks)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "comm_create_local: invalid argument");
    gt4mi_comm* c = new gt4mi_comm;  // no RCCL communicator behind it: plans on it exchange through the direct transport only
    c->nranks = nranks;
    c->rank = rank;
    *comm = c;
    return GT4MI_OK;
}

int gt4mi_comm_info(gt4mi_comm* comm, int* nranks, int* rank, int* device) {
    if (comm == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "comm_info: null communicator");
    if (comm->comm == nullptr) {  // gt4mi_comm_create_local: what the caller said, and the current device
        if (nranks) *nranks = comm->nranks;
        if (rank) *rank = comm->rank;
        if (device) GT4MI_HIP_CHECK(hipGetDevice(device));
        return GT4MI_OK;
    }
    gt4mi::RcclApi& api = gt4mi::rccl();
    // what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), not what the
    // caller passed to gt4mi_comm_create
    if (nranks) {
        if (!api.CommCount) return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "comm_info: ncclCommCount not available");
        GT4MI_RCCL_CHECK(api.CommCount(comm->comm, nranks));
    }
    if (rank) {
        if (!api.CommUserRank) return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "comm_info: ncclCommUserRank not available");
        GT4MI_RCCL_CHECK(api.CommUserRank(comm->comm, rank));
    }
    if (device) {
        if (!api.CommCuDevice) return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "comm_info: ncclCommCuDevice not available");
        GT4MI_RCCL_CHECK(api.CommCuDevice(comm->comm, device));
    }
    return GT4MI_OK;
}

int gt4mi_comm_destroy(gt4mi_comm* comm) {
    if (comm == nullptr) return GT4MI_OK;
    if (comm->comm) gt4mi::rccl().CommDestroy(comm->comm);
    delete comm;
    return GT4MI_OK;
}

int gt4mi_halo_plan_create(gt4mi_comm* comm, int elem_size, const gt4mi_halo_msg* sends, int nsends,
                           const gt4mi_halo_msg* recvs, int nrecvs, gt4mi_halo_plan** plan) {
    if (comm == nullptr || plan == nullptr || (nsends > 0 && sends == nullptr) || (nrecvs > 0 && recvs == nullptr))
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_create: null argument");
    if (elem_size != 4 && elem_size != 8)
        return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "halo_plan_create: element size %d", elem_size);
    gt4mi_halo_plan* p = new gt4mi_halo_plan;
    p->comm = comm;
    p->elem_size = elem_size;
    auto add = [&](const gt4mi_halo_msg& m, bool is_send) -> int {
        if (m.phase < 0 || m.phase > 1 || m.peer < 0 || m.peer >= comm->nranks)
            return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_create: bad phase/peer");
        gt4mi_halo_plan::Msg x;
        x.peer = m.peer;
        size_t n = 1;
        for (int a = 0; a < 3; ++a) {
            if (m.extent[a] < 0 || m.lo[a] < 0) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_create: bad box");
            x.lo[a] = m.lo[a];
            x.ext[a] = m.extent[a];
            n *= (size_t)m.extent[a];
        }
        x.bytes = n * (size_t)elem_size;
        x.buffer = nullptr;
        if (x.bytes) GT4MI_HIP_CHECK(hipMalloc(&x.buffer, x.bytes));
        (is_send ? p->sends : p->recvs)[m.phase].push_back(x);
        return GT4MI_OK;
    };
    int rc = GT4MI_OK;
    for (int i = 0; i < nsends && rc == GT4MI_OK; ++i) rc = add(sends[i], true);
    for (int i = 0; i < nrecvs && rc == GT4MI_OK; ++i) rc = add(recvs[i], false);
    if (rc == GT4MI_OK) {
        // Normal priority on purpose: measured on MI355X (1-rank self-loop rehearsal, 512x64x512 per
        // step) a highest-priority side stream made the step 3x SLOWER (0.267 ms vs 0.088 ms) and a
        // lowest-priority one 1.6x slower.
        if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess)
            rc = gt4mi::fail(GT4MI_ERR_HIP, "halo_plan_create: hipStreamCreate failed");
    }
    if (rc == GT4MI_OK && (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess))
        rc = gt4mi::fail(GT4MI_ERR_HIP, "halo_plan_create: hipEventCreate failed");
    if (rc == GT4MI_OK) {
        void* words = nullptr;
        if (hipMalloc(&words, 256) != hipSuccess || hipMemset(words, 0, 256) != hipSuccess)
            rc = gt4mi::fail(GT4MI_ERR_HIP, "halo_plan_create: hipMalloc failed");
        p->probe = static_cast<unsigned*>(words);
        p->edge_words = words ? reinterpret_cast<uint32_t*>(words) + 32 : nullptr;  // (the second half of the 256 bytes)
    }
    if (rc != GT4MI_OK) {
        gt4mi_halo_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return GT4MI_OK;
}

int gt4mi_halo_plan_set_option(gt4mi_halo_plan* plan, int option, int value) {
    if (plan == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: null plan");
    switch (option) {
        case GT4MI_PLAN_SCHEDULE:
            if (value < -1 || value > GT4MI_SCHEDULE_INLINE) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: schedule %d", value);
            plan->schedule = value;
            return GT4MI_OK;
        case GT4MI_PLAN_EDGE_COLUMNS:
            if (value < -1 || value > 4096) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: %d edge columns", value);
            plan->edge_columns = value;
            return GT4MI_OK;
        case GT4MI_PLAN_DEFER_JOIN:
            if (value != 0 && value != 1) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: defer_join %d", value);
            plan->defer_join = value;
            return GT4MI_OK;
        case GT4MI_PLAN_INTERIOR_WG_PER_CU:
            if (value < -1 || value > 16) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: %d workgroups per CU", value);
            plan->interior_wg_per_cu = value;
            return GT4MI_OK;
        case GT4MI_PLAN_TRANSPORT:
            if (value != GT4MI_TRANSPORT_RCCL && value != GT4MI_TRANSPORT_DIRECT)
                return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: transport %d", value);
            if (value == GT4MI_TRANSPORT_DIRECT) {
                if (!plan->direct.prepared)
                    return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: prepare and connect the direct transport first "
                                                                   "(gt4mi_halo_plan_direct_prepare / _connect)");
                for (int ph = 0; ph < 2; ++ph) {
                    for (size_t m = 0; m < plan->sends[ph].size(); ++m)
                        if (!plan->direct.send_to[ph][m] || !plan->direct.signal_arrived[ph][m])
                            return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: send %d of phase %d is not connected", (int)m, ph);
                    for (size_t m = 0; m < plan->recvs[ph].size(); ++m)
                        if (!plan->direct.signal_consumed[ph][m])
                            return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: receive %d of phase %d is not connected", (int)m, ph);
                }
            }
            plan->transport = value;
            return GT4MI_OK;
        case GT4MI_PLAN_DIRECT_TIMEOUT_MS:
            if (value < 0) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: a timeout of %d ms", value);
            plan->direct.timeout_ms = value;
            return GT4MI_OK;
        case GT4MI_PLAN_DIRECT_FENCED:
            if (value != 0 && value != 1) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: direct_fenced %d", value);
            plan->direct.fenced = value;
            return GT4MI_OK;
    }
    return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_set_option: unknown option %d", option);
}

int gt4mi_halo_plan_concurrent(gt4mi_halo_plan* plan) {
    if (plan == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_concurrent: null plan");
    return plan->probed ? (plan->concurrent ? 1 : 0) : 2;
}

int gt4mi_halo_plan_direct_prepare(gt4mi_halo_plan* plan, gt4mi_direct_info* info) {
    if (plan == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_direct_prepare: null plan");
    return gt4mi::direct_prepare(plan, info);
}

int gt4mi_halo_plan_direct_layout(gt4mi_halo_plan* plan, int phase, int is_send, int index, int64_t* pool_offset, int* flag_index) {
    if (plan == nullptr || !plan->direct.prepared) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_direct_layout: not prepared");
    if (phase < 0 || phase > 1) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_direct_layout: phase %d", phase);
    const size_t n = is_send ? plan->sends[phase].size() : plan->recvs[phase].size();
    if (index < 0 || (size_t)index >= n) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_direct_layout: message %d of %d", index, (int)n);
    if (pool_offset) *pool_offset = is_send ? -1 : (int64_t)plan->direct.recv_offset[phase][index];
    if (flag_index) *flag_index = gt4mi::direct_index(plan, is_send != 0, phase, index);
    return GT4MI_OK;
}

int gt4mi_halo_plan_direct_connect(gt4mi_halo_plan* plan, int phase, int is_send, int index, const gt4mi_direct_info* peer,
                                   int64_t peer_pool_offset, int peer_flag_index) {
    if (plan == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_direct_connect: null plan");
    return gt4mi::direct_connect(plan, phase, is_send, index, peer, peer_pool_offset, peer_flag_index);
}

int gt4mi_halo_plan_direct_status(gt4mi_halo_plan* plan, int* timed_out, unsigned* exchanges) {
    if (plan == nullptr || !plan->direct.prepared) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_plan_direct_status: not prepared");
    GT4MI_HIP_CHECK(hipDeviceSynchronize());  // every exchange started so far has either completed or given up
    const uint32_t word = __atomic_load_n(plan->direct.error, __ATOMIC_RELAXED);  // (host memory the device writes)
    if (timed_out) *timed_out = (int)word;
    if (exchanges) *exchanges = plan->direct.step;
    return GT4MI_OK;
}

int gt4mi_halo_plan_destroy(gt4mi_halo_plan* plan) {
    if (plan == nullptr) return GT4MI_OK;
    gt4mi::direct_release(plan);  // (the receive buffers of a prepared plan live in its pool)
    for (int ph = 0; ph < 2; ++ph) {
        for (auto& m : plan->sends[ph]) if (m.buffer) (void)hipFree(m.buffer);
        for (auto& m : plan->recvs[ph]) if (m.buffer) (void)hipFree(m.buffer);
    }
    if (plan->probe) (void)hipFree(plan->probe);
    if (plan->ready) (void)hipEventDestroy(plan->ready);
    if (plan->done) (void)hipEventDestroy(plan->done);
    if (plan->stream) (void)hipStreamDestroy(plan->stream);
    delete plan;
    return GT4MI_OK;
}

int gt4mi_halo_exchange(gt4mi_halo_plan* plan, const gt4mi_field* field, void* stream) {
    if (plan == nullptr || field == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_exchange: null argument");
    return gt4mi::halo_exchange_on(plan, field, static_cast<hipStream_t>(stream));
}

int gt4mi_halo_exchange_fork(gt4mi_halo_plan* plan, void* main_stream) {
    if (plan == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_exchange_fork: null plan");
    // first use with this stream: make sure the side stream does not share its hardware queue (one-off,
    // synchronising probe) -- otherwise the "overlapped" exchange simply queues behind the interior kernel
    if (int rc = gt4mi::ensure_concurrent_stream(plan, static_cast<hipStream_t>(main_stream))) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->ready, static_cast<hipStream_t>(main_stream)));
    plan->forked = true;
    return GT4MI_OK;
}

int gt4mi_halo_exchange_begin(gt4mi_halo_plan* plan, const gt4mi_field* field, void* main_stream) {
    if (plan == nullptr || field == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_exchange_begin: null argument");
    if (!plan->forked)
        if (int rc = gt4mi::ensure_concurrent_stream(plan, static_cast<hipStream_t>(main_stream))) return rc;
    if (!plan->forked) GT4MI_HIP_CHECK(hipEventRecord(plan->ready, static_cast<hipStream_t>(main_stream)));
    plan->forked = false;
    GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
    if (int rc = gt4mi::halo_exchange_on(plan, field, plan->stream)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
    plan->primed = true;
    return GT4MI_OK;
}

int gt4mi_halo_exchange_end(gt4mi_halo_plan* plan, void* main_stream) {
    if (plan == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "halo_exchange_end: null plan");
    if (int rc = gt4mi::direct_failed(plan)) return rc;  // a wait of an EARLIER exchange ran out of time: say so now
    if (!plan->done_recorded) return GT4MI_OK;  // nothing was ever put in flight on the side stream
    GT4MI_HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(main_stream), plan->done, 0));
    return GT4MI_OK;
}

int gt4mi_dist_lap5_f64(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                        const gt4mi_field* out, int variant, int sides, void* main_stream) {
    return dist_lap5<double, double>(plan, domain, inp, out, variant, sides, main_stream);
}

int gt4mi_dist_lap5_query(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int sides,
                          int* edge_units) {
    if (plan == nullptr || inp == nullptr || out == nullptr || domain == nullptr || edge_units == nullptr)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_lap5_query: null argument");
    gt4mi::EdgeFaces g;
    gt4mi::EdgeCopies cp;
    int phase = 0;
    bool ok = false;
    int rc;
    if (plan->elem_size == 8) {
        gt4mi::View<double> vi, vo;
        rc = gt4mi::lap5_edge_prepare<double>(plan, domain, inp, out, sides, &vi, &vo, &g, &cp, &phase, &ok);
    } else {
        gt4mi::View<float> vi, vo;
        rc = gt4mi::lap5_edge_prepare<float>(plan, domain, inp, out, sides, &vi, &vo, &g, &cp, &phase, &ok);
    }
    *edge_units = ok ? 1 : 0;
    return rc;
}

int gt4mi_dist_lap5_f32(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                        const gt4mi_field* out, int variant, int flags, int sides, void* main_stream) {
    if (flags & GT4MI_LAP_LITERAL_F32) return dist_lap5<float, float>(plan, domain, inp, out, variant, sides, main_stream);
    return dist_lap5<float, double>(plan, domain, inp, out, variant, sides, main_stream);
}

int gt4mi_dist_lap5_f64_wide(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                             const gt4mi_field* out, int variant, int sides, int halo, int phase, void* main_stream) {
    if (plan == nullptr || inp == nullptr || out == nullptr || domain == nullptr)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_lap5_wide: null argument");
    if (halo < 1 || phase < 0 || phase >= halo)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_lap5_wide: need halo >= 1 and 0 <= phase < halo");
    if (!plan->primed)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT,
                           "dist_lap5_wide: the ghost cells of the first input were never exchanged "
                           "(call gt4mi_halo_exchange_begin on it once before the first step)");
    hipStream_t ms = static_cast<hipStream_t>(main_stream);
    const int64_t di = domain[0], dj = domain[1], dk = domain[2], H = halo;
    const bool w = sides & 1, e = sides & 2, s = sides & 4, n = sides & 8;
    if ((w || e) && di < 2 * H) return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "dist_lap5_wide: local I extent smaller than 2*halo");
    if ((s || n) && dj < 2 * H) return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "dist_lap5_wide: local J extent smaller than 2*halo");
    // region [i0, i1) x [j0, j1) relative to the local compute-domain origin
    auto run = [&](int64_t i0, int64_t i1, int64_t j0, int64_t j1) -> int {
        if (i1 <= i0 || j1 <= j0 || dk <= 0) return GT4MI_OK;
        gt4mi_field a = *inp, b = *out;
        a.origin[0] += i0; a.origin[1] += j0;
        b.origin[0] += i0; b.origin[1] += j0;
        const int64_t d[3] = {i1 - i0, j1 - j0, dk};
        return gt4mi::lap5_run<double, double>(d, &a, &b, variant, ms);
    };
    if (phase == 0) {
        if (!(plan->probed && plan->probed_main == ms)) {
            // the probe synchronises: keep the exchange in flight ordered before it
            GT4MI_HIP_CHECK(hipStreamWaitEvent(ms, plan->done, 0));
            if (int rc = gt4mi::ensure_concurrent_stream(plan, ms)) return rc;
            GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
        }
        // join the exchange that delivered `inp`'s ghost cells (started `halo` steps ago)
        GT4MI_HIP_CHECK(hipStreamWaitEvent(ms, plan->done, 0));
    }
    const int64_t ext = H - 1 - phase;  // how far this step still reaches into the ghost region
    if (ext > 0) {
        // redundant-compute step: one launch over the domain grown by `ext` towards every neighbour;
        // its ghost results are valid inputs for the next step, no communication
        return run(w ? -ext : 0, di + (e ? ext : 0), s ? -ext : 0, dj + (n ? ext : 0));
    }
    // last step of the cycle: `out`'s faces (H deep) are what the neighbours need next
    const int64_t lo_i = w ? H : 0, hi_i = e ? H : 0, lo_j = s ? H : 0, hi_j = n ? H : 0;
    {  // the H-deep ring of `out` in ONE launch (lap5_ring.hip.h)
        const int outer[4] = {0, 0, 0, 0};
        const int inner[4] = {(int)lo_i, (int)hi_i, (int)lo_j, (int)hi_j};
        if (int rc = gt4mi::lap5_ring_run<double, double>(domain, inp, out, variant, outer, inner, ms)) return rc;
    }
    // pack on the main stream (before the interior kernel floods the CUs), then fork
    if (int rc = gt4mi::halo_pack_first(plan, out, ms)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
    GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
    // interior on the main stream, RCCL send/recv + unpack of `out`'s ghost cells next to it; nobody
    // waits for them until phase 0 of the next cycle (where `out` is the input)
    if (int rc = run(lo_i, di - hi_i, lo_j, dj - hi_j)) return rc;
    if (int rc = gt4mi::halo_exchange_on(plan, out, plan->stream, /*first_pack_done=*/true)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
    return GT4MI_OK;
}

int gt4mi_dist_lap5_f64_skewed(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* field_a,
                               const gt4mi_field* field_b, int variant, int sides, int halo, void* main_stream) {
    if (plan == nullptr || field_a == nullptr || field_b == nullptr || domain == nullptr)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_lap5_skewed: null argument");
    if (halo < 1) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "dist_lap5_skewed: need halo >= 1");
    if (!plan->primed)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT,
                           "dist_lap5_skewed: the ghost cells of the first input were never exchanged "
                           "(call gt4mi_halo_exchange_begin on it once before the first cycle)");
    hipStream_t ms = static_cast<hipStream_t>(main_stream);
    const int64_t di = domain[0], dj = domain[1], dk = domain[2];
    const int H = halo;
    const bool w = sides & 1, e = sides & 2, s = sides & 4, n = sides & 8;
    // the band of step 1 reaches 2H - 1 points into the domain from every side that has a neighbour
    if ((w || e) && di < (int64_t)(2 * H - 1) * ((w ? 1 : 0) + (e ? 1 : 0)))
        return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "dist_lap5_skewed: local I extent too small for a ghost depth of %d", H);
    if ((s || n) && dj < (int64_t)(2 * H - 1) * ((s ? 1 : 0) + (n ? 1 : 0)))
        return gt4mi::fail(GT4MI_ERR_UNSUPPORTED, "dist_lap5_skewed: local J extent too small for a ghost depth of %d", H);
    if (!(plan->probed && plan->probed_main == ms)) {
        // the probe synchronises: keep the exchange in flight ordered before it
        GT4MI_HIP_CHECK(hipStreamWaitEvent(ms, plan->done, 0));
        if (int rc = gt4mi::ensure_concurrent_stream(plan, ms)) return rc;
        GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
    }
    // join the exchange that delivered field_a's ghost cells (started by the previous cycle)
    GT4MI_HIP_CHECK(hipStreamWaitEvent(ms, plan->done, 0));
    auto src_of = [&](int step) { return (step % 2 == 1) ? field_a : field_b; };  // step 1 reads a, writes b
    auto dst_of = [&](int step) { return (step % 2 == 1) ? field_b : field_a; };
    // 1. the bands, outermost first: step st on [-(H - st), 2H - st) points from every side with a neighbour
    for (int st = 1; st <= H; ++st) {
        const int g = H - st, d = 2 * H - st;
        const int outer[4] = {w ? g : 0, e ? g : 0, s ? g : 0, n ? g : 0};
        const int inner[4] = {w ? d : 0, e ? d : 0, s ? d : 0, n ? d : 0};
        if (int rc = gt4mi::lap5_ring_run<double, double>(domain, src_of(st), dst_of(st), variant, outer, inner, ms)) return rc;
    }
    // 2. the H-deep faces of the result are final: pack them on the main stream (before the interior kernels flood the
    //    device), then the side stream sends / receives / unpacks next to ALL H interior kernels
    const gt4mi_field* result = dst_of(H);
    // (chain schedule: the pack runs on the side stream as well, next to the first interior kernel)
    const bool pack_on_side = gt4mi::plan_schedule(plan, GT4MI_SCHEDULE_JOIN) == GT4MI_SCHEDULE_CHAIN;
    if (!pack_on_side)
        if (int rc = gt4mi::halo_pack_first(plan, result, ms)) return rc;
    GT4MI_HIP_CHECK(hipEventRecord(plan->ready, ms));
    GT4MI_HIP_CHECK(hipStreamWaitEvent(plan->stream, plan->ready, 0));
    // 3. the interiors: step st on the domain shrunk by 2H - st
    for (int st = 1; st <= H; ++st) {
        const int64_t d = 2 * H - st;
        const int64_t i0 = w ? d : 0, i1 = di - (e ? d : 0), j0 = s ? d : 0, j1 = dj - (n ? d : 0);
        if (i1 > i0 && j1 > j0 && dk > 0) {
            gt4mi_field a = *src_of(st), b = *dst_of(st);
            a.origin[0] += i0; a.origin[1] += j0;
            b.origin[0] += i0; b.origin[1] += j0;
            const int64_t sub[3] = {i1 - i0, j1 - j0, dk};
            gt4mi::ScopedLaunchLds throttle(gt4mi::lds_for_workgroups_per_cu(gt4mi::plan_interior_wg_per_cu(plan, 0)));
            if (int rc = gt4mi::lap5_run<double, double>(sub, &a, &b, variant, ms)) return rc;
        }
        if (st == 1) {  // enqueued after the first interior launch so that the device has work while the host talks to RCCL
            if (int rc = gt4mi::halo_exchange_on(plan, result, plan->stream, /*first_pack_done=*/!pack_on_side)) return rc;
            GT4MI_HIP_CHECK(hipEventRecord(plan->done, plan->stream));
        plan->done_recorded = true;
        }
    }
    return GT4MI_OK;
}

int gt4mi_dist_lap5_f64_pipelined(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                                  const gt4mi_field* out, int variant, int sides, void* main_stream) {
    return gt4mi_dist_lap5_f64_wide(plan, domain, inp, out, variant, sides, 1, 0, main_stream);
}


int gt4mi_dist_hdiff_f64(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* in_field,
                         const gt4mi_field* out_field, const gt4mi_field* coeff, double coeff_scalar, int flags, int sides,
                         void* main_stream) {
    return dist_hdiff<double>(plan, domain, in_field, out_field, coeff, coeff_scalar, flags, sides, main_stream);
}

int gt4mi_dist_hdiff_f32(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* in_field,
                         const gt4mi_field* out_field, const gt4mi_field* coeff, double coeff_scalar, int flags, int sides,
                         void* main_stream) {
    return dist_hdiff<float>(plan, domain, in_field, out_field, coeff, coeff_scalar, flags, sides, main_stream);
}

int gt4mi_stream_copy(const void* src, void* dst, size_t nbytes, void* stream) {
    if (src == nullptr || dst == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "stream_copy: null pointer");
    if (nbytes % 16 != 0 || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 != 0)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "stream_copy: pointers and size must be multiples of 16 bytes");
    const size_t nvec = nbytes / 16;
    if (nvec == 0) return GT4MI_OK;
    // best of the variants in `microbench copy` (profiles/r1_microbench_*.log): one 16-byte vector per
    // thread, no grid-stride loop, non-temporal stores -- 6.23 TB/s on MI355X
    constexpr int UNROLL = 1;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
    hipLaunchKernelGGL((gt4mi::stream_copy_kernel<UNROLL, true>), dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const gt4mi::u32x4*>(src),
                       static_cast<gt4mi::u32x4*>(dst), nvec);
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

int gt4mi_memory_write_probe(void* a, void* b, size_t bytes, int iterations, void* stream, double* gbs) {
    return gt4mi::memory_write_probe(a, b, bytes, iterations, static_cast<hipStream_t>(stream), gbs);
}

// ---- run-time compiled stencils (generic executor) -------------------------------------------------

int gt4mi_rtc_compile(const char* source, const char* name, const char* const* options, int n_options,
                      void** code, size_t* code_size, char* log, size_t log_size) {
    return gt4mi::rtc_compile(source, name, options, n_options, code, code_size, log, log_size);
}

int gt4mi_rtc_free(void* code) {
    free(code);
    return GT4MI_OK;
}

int gt4mi_module_load(const void* code, gt4mi_module** module) {
    if (code == nullptr || module == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "module_load: null argument");
    auto* m = new gt4mi_module();
    hipError_t e = hipModuleLoadData(&m->module, code);
    if (e != hipSuccess) {
        delete m;
        return gt4mi::fail(GT4MI_ERR_HIP, "hipModuleLoadData failed: %s", hipGetErrorString(e));
    }
    *module = m;
    return GT4MI_OK;
}

int gt4mi_module_unload(gt4mi_module* module) {
    if (module == nullptr) return GT4MI_OK;
    hipError_t e = module->module ? hipModuleUnload(module->module) : hipSuccess;
    delete module;
    if (e != hipSuccess) return gt4mi::fail(GT4MI_ERR_HIP, "hipModuleUnload failed: %s", hipGetErrorString(e));
    return GT4MI_OK;
}

int gt4mi_module_function(gt4mi_module* module, const char* name, void** function) {
    if (module == nullptr || name == nullptr || function == nullptr)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "module_function: null argument");
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleGetFunction(&fn, module->module, name);
    if (e != hipSuccess)
        return gt4mi::fail(GT4MI_ERR_HIP, "hipModuleGetFunction(%s) failed: %s", name, hipGetErrorString(e));
    *function = fn;
    return GT4MI_OK;
}

int gt4mi_function_info(void* function, int* registers, int* scratch_bytes, int* lds_bytes) {
    if (function == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "function_info: null function");
    hipFunction_t fn = static_cast<hipFunction_t>(function);
    int v = 0;
    if (registers) {
        GT4MI_HIP_CHECK(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn));
        *registers = v;
    }
    if (scratch_bytes) {
        GT4MI_HIP_CHECK(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn));
        *scratch_bytes = v;
    }
    if (lds_bytes) {
        GT4MI_HIP_CHECK(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fn));
        *lds_bytes = v;
    }
    return GT4MI_OK;
}

int gt4mi_launch(void* function, const uint32_t grid[3], const uint32_t block[3], const void* args,
                 size_t args_size, void* stream, gt4mi_exec_info* info) {
    Timer timer(info, stream);
    if (function == nullptr || grid == nullptr || block == nullptr || (args == nullptr && args_size != 0))
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "launch: null argument");
    if (grid[0] == 0 || grid[1] == 0 || grid[2] == 0) return GT4MI_OK;  // empty iteration space
    if ((size_t)block[0] * block[1] * block[2] > 1024 || block[0] * block[1] * block[2] == 0)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "launch: workgroup of %u x %u x %u threads", block[0],
                           block[1], block[2]);
    if (grid[1] > 65535u || grid[2] > 65535u)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "launch: grid %u x %u x %u exceeds 65535 in y or z", grid[0],
                           grid[1], grid[2]);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args), HIP_LAUNCH_PARAM_BUFFER_SIZE,
                     &args_size, HIP_LAUNCH_PARAM_END};
    GT4MI_HIP_CHECK(hipModuleLaunchKernel(static_cast<hipFunction_t>(function), grid[0], grid[1], grid[2], block[0],
                                          block[1], block[2], 0, static_cast<hipStream_t>(stream), nullptr, extra));
    return GT4MI_OK;
}

int gt4mi_launch_batch(int n, void* const* functions, const uint32_t* grids, const uint32_t* blocks,
                       const void* const* args, size_t args_size, void* stream, gt4mi_exec_info* info) {
    Timer timer(info, stream);
    if (n < 0 || (n > 0 && (functions == nullptr || grids == nullptr || blocks == nullptr || args == nullptr)))
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "launch_batch: null argument");
    for (int l = 0; l < n; ++l) {
        const int rc = gt4mi_launch(functions[l], grids + 3 * l, blocks + 3 * l, args[l], args_size, stream, nullptr);
        if (rc != GT4MI_OK) return rc;
    }
    return GT4MI_OK;
}

}  // extern "C"
