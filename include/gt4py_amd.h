/*
 * gt4py_amd.h -- C ABI of the MI355X-native stencil-execution library (libgt4py_amd.so).
 *
 * This is the drop-in boundary for the hot path of gt4py.cartesian.  In the
 * reference every compiled backend exposes ONE native entry per stencil, generated at JIT time:
 *
 *     run_computation(std::array<uint_t,3> domain,
 *                     {py::buffer|py::object field, std::array<int_t,ndim> field_origin}...,
 *                     scalars by value..., py::object exec_info)
 *         -- /root/reference/src/gt4py/cartesian/backend/gtc_common.py:65-103  (bindings template)
 *         -- /root/reference/src/gt4py/cartesian/backend/gtcpp_backend.py:77-106 (argument marshalling)
 *         -- /root/reference/src/gt4py/cartesian/backend/gtc_common.py:30-62  (buffer -> SID, origin shift)
 *
 * The functions below are the same entry, one per hand-written kernel family, with the pybind11
 * objects replaced by plain pointers and sizes so that they can be bound from ctypes / cffi / any
 * FFI.  A field is described exactly by what `pybuffer_to_sid` extracts from the Python buffer
 * (pointer, shape, byte strides) plus the per-field origin the generated wrapper passes next to it.
 *
 * All pointers are DEVICE pointers (HIP, gfx950).  No function allocates or frees caller memory.
 * Every function returns 0 on success or a negative gt4mi_status; the message of the last failure
 * on the calling thread is available from gt4mi_last_error().  Launches are asynchronous on
 * `stream` (a hipStream_t passed as void*; NULL = the default stream); the caller synchronises
 * (the reference synchronises the device after every gt:gpu call unless device_sync=False --
 * backend/gtc_common.py:288-296; the Python host code does the same through gt4mi_stream_sync).
 */
#ifndef GT4PY_AMD_H
#define GT4PY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GT4MI_ABI_VERSION 7 /* 7: gt4mi_memory_write_probe (which memory group an allocation lives in); 6: GT4MI_PLAN_DIRECT_FENCED (release / acquire fences around the flags of the direct transport); 5: GT4MI_ERR_TIMEOUT (a direct-transport wait that runs out fails the plan, hard), GT4MI_PLAN_DIRECT_TIMEOUT_MS; 4: gt4mi_dist_lap5_f32, the direct transport (gt4mi_halo_plan_direct_*, GT4MI_PLAN_TRANSPORT), gt4mi_comm_create_local, schedules 2-4 */

typedef enum gt4mi_status {
    GT4MI_OK = 0,
    GT4MI_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad enum, negative size ...              */
    GT4MI_ERR_OUT_OF_BOUNDS = -2,    /* origin/domain/halo do not fit in the field's shape     */
    GT4MI_ERR_UNSUPPORTED = -3,      /* combination not implemented by any kernel              */
    GT4MI_ERR_HIP = -4,              /* a HIP runtime call failed; see gt4mi_last_error()      */
    GT4MI_ERR_TIMEOUT = -5           /* direct transport: a neighbour never arrived; the plan has failed for good */
} gt4mi_status;

/* One stencil field argument.
 * Replaces the (py::buffer, origin) pair of run_computation (gtc_common.py:80-82, 30-62). */
typedef struct gt4mi_field {
    void* data;        /* device address of element [0,0,0]                                  */
    int64_t shape[3];  /* extent along I, J, K                                               */
    int64_t stride[3]; /* BYTE strides along I, J, K (any layout; I-contiguous is the fast one) */
    int64_t origin[3]; /* index of the first compute-domain point (the `_origin_[name]` entry) */
} gt4mi_field;

/* Optional timestamps, seconds of the monotonic clock (CLOCK_MONOTONIC = Python's time.perf_counter()).  May be NULL.
 *   run_cpp_*  the counterpart of exec_info["run_cpp_start_time"/"..end_time"] (gtc_common.py:83-99): they bracket
 *              the native call.  Launches are asynchronous here, so this is the time to ENQUEUE the work.
 *   run_hip_*  the time the work spent on the device, from a hipEvent pair on the launch stream: the interval ends
 *              when the host saw the stop event complete and is as long as the events measured.  Passing a
 *              non-NULL struct therefore makes the call wait for its own kernels (and only then).
 * Invariant: run_cpp_start_time <= run_hip_start_time <= run_hip_end_time. */
typedef struct gt4mi_exec_info {
    double run_cpp_start_time;
    double run_cpp_end_time;
    double run_hip_start_time;
    double run_hip_end_time;
} gt4mi_exec_info;

/* ---- library ------------------------------------------------------------------------------- */
int gt4mi_abi_version(void);
const char* gt4mi_last_error(void);
/* Writes a NUL-terminated description of the current HIP device (name, arch, CUs) into buf. */
int gt4mi_device_info(char* buf, size_t buflen);
int gt4mi_stream_sync(void* stream);

/* ---- 5-point star stencils (single statement, PARALLEL, interval(...)) ------------------------
 * Replaces run_computation of the stencils generated from
 *   variant 0: examples/lap_cartesian_vs_next.ipynb cell 7
 *              out = -4.0*inp[0,0,0] + inp[-1,0,0] + inp[1,0,0] + inp[0,-1,0] + inp[0,1,0]
 *   variant 1: docs/user/cartesian/index.rst:24-28
 *              out = -4.*inp + (inp[I+1] + inp[I-1] + inp[J+1] + inp[J-1])
 *   variant 2: tests/.../multi_feature_tests/test_suites.py:214 (Laplacian of horizontal diffusion)
 *              out = 4.0*inp - (inp[1,0,0] + inp[-1,0,0] + inp[0,1,0] + inp[0,-1,0])
 *   variant 3: tests/.../feature_tests/test_call_interface.py:159-164
 *              out = 0.25*(inp[0,1,0] + inp[0,-1,0] + inp[1,0,0] + inp[-1,0,0])
 * Expression trees are evaluated exactly as parsed (left-assoc, one rounding per op, no FMA).
 * `inp` needs a halo of 1 in I and J around the compute domain (field_info boundary
 * ((1,1),(1,1),(0,0)), module_generator.py:56-106); `inp` and `out` must not overlap
 * (gtir_to_oir.py:19-46 rejects such stencils). */
enum { GT4MI_LAP_NOTEBOOK = 0, GT4MI_LAP_DOCS = 1, GT4MI_LAP_SUITE = 2, GT4MI_LAP_AVG = 3 };
/* flags (f32 entry only): GT4MI_LAP_LITERAL_F32 = float literals typed float32
 * (literal_float_precision=32, frontend/gtscript_frontend.py:1250-1259): all arithmetic in float.
 * Default: literals are float64, so float fields are widened, computed in double and rounded once
 * on store (gtc/passes/gtir_upcaster.py:43-143). */
enum { GT4MI_LAP_LITERAL_F32 = 1 };

int gt4mi_lap5_f64(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out,
                   int variant, int flags, void* stream, gt4mi_exec_info* info);
int gt4mi_lap5_f32(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out,
                   int variant, int flags, void* stream, gt4mi_exec_info* info);

/* ---- horizontal diffusion ----------------------------------------------------------------------
 * Replaces run_computation of `horizontal_diffusion` (flux limiter) and
 * `simple_horizontal_diffusion` / TestHorizontalDiffusion (no limiter):
 *   tests/.../multi_feature_tests/stencil_definitions.py:316-328, :206-216; test_suites.py:212-220.
 * `in_field` needs a halo of 2 in I and J.  The diffusion coefficient is a field (coeff != NULL)
 * or a scalar parameter (coeff == NULL, value in coeff_scalar; for the f32 entry the scalar is
 * first rounded to float when GT4MI_HDIFF_COEFF_F32 is set, i.e. the parameter was declared
 * float32).  flags:
 *   GT4MI_HDIFF_LIMITER       apply the flux limiter (ternary -> select, npir_codegen.py:225)
 *   GT4MI_HDIFF_INTERNAL_F32  (f32 entry only) literals typed float32 (literal_float_precision=32):
 *                             all arithmetic in float.  Default follows the reference default
 *                             (float64 literals => lap/flx/fly in double, gtir_upcaster.py:43-143). */
enum { GT4MI_HDIFF_LIMITER = 1, GT4MI_HDIFF_INTERNAL_F32 = 2, GT4MI_HDIFF_COEFF_F32 = 4 };

int gt4mi_hdiff_f64(const int64_t domain[3], const gt4mi_field* in_field,
                    const gt4mi_field* out_field, const gt4mi_field* coeff, double coeff_scalar,
                    int flags, void* stream, gt4mi_exec_info* info);
int gt4mi_hdiff_f32(const int64_t domain[3], const gt4mi_field* in_field,
                    const gt4mi_field* out_field, const gt4mi_field* coeff, double coeff_scalar,
                    int flags, void* stream, gt4mi_exec_info* info);

/* The same stencils on a RING of the compute domain only (NEW: the boundary part of an IJ-decomposed apply, SURVEY.md
 * section 8e; one launch for up to four boxes).  widths / inner / outer are {low I, high I, low J, high J}.
 *   gt4mi_hdiff_ring_*: the points less than widths[side] away from a side of the domain (0 = that side has no ring);
 *                       same arguments, bounds and alias rules as gt4mi_hdiff_*.
 *   gt4mi_lap5_ring_*:  (the domain grown by outer[side]) minus (the domain shrunk by inner[side]); `inp` must be readable
 *                       one point beyond the grown domain.  outer > 0 is what communication-avoiding time stepping
 *                       computes redundantly inside its ghost region (gt4mi_dist_lap5_f64_skewed). */
int gt4mi_hdiff_ring_f64(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                         const gt4mi_field* coeff, double coeff_scalar, int flags, const int widths[4], void* stream,
                         gt4mi_exec_info* info);
int gt4mi_hdiff_ring_f32(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                         const gt4mi_field* coeff, double coeff_scalar, int flags, const int widths[4], void* stream,
                         gt4mi_exec_info* info);
int gt4mi_lap5_ring_f64(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant, int flags,
                        const int outer[4], const int inner[4], void* stream, gt4mi_exec_info* info);
int gt4mi_lap5_ring_f32(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant, int flags,
                        const int outer[4], const int inner[4], void* stream, gt4mi_exec_info* info);

/* ---- vertical tridiagonal (Thomas) solve --------------------------------------------------------
 * Replaces run_computation of `tridiagonal_solver` (stencil_definitions.py:219-232):
 * FORWARD sweep rewrites sup and rhs IN PLACE (they are READ_WRITE API fields), BACKWARD sweep
 * writes out.  domain[2] must be >= 2 (min_sequential_axis_size, gtir_k_boundary.py:78-109). */
int gt4mi_tridiag_f64(const int64_t domain[3], const gt4mi_field* inf, const gt4mi_field* diag,
                      const gt4mi_field* sup, const gt4mi_field* rhs, const gt4mi_field* out,
                      void* stream, gt4mi_exec_info* info);
int gt4mi_tridiag_f32(const int64_t domain[3], const gt4mi_field* inf, const gt4mi_field* diag,
                      const gt4mi_field* sup, const gt4mi_field* rhs, const gt4mi_field* out,
                      void* stream, gt4mi_exec_info* info);

/* ---- halo pack / unpack (NEW: the reference has no multi-device path) -----
 * Copies the box [lo, lo+extent) of `field` (indices relative to element [0,0,0], NOT to the
 * origin) to / from a dense buffer laid out I-fastest, then J, then K.  elem_size is 4 or 8. */
int gt4mi_halo_pack(const gt4mi_field* field, const int64_t lo[3], const int64_t extent[3],
                    void* buffer, int elem_size, void* stream);
int gt4mi_halo_unpack(const gt4mi_field* field, const int64_t lo[3], const int64_t extent[3],
                      const void* buffer, int elem_size, void* stream);

/* ---- multi-GPU: RCCL halo exchange driven from native code (NEW, no reference counterpart) --------
 * One process per GPU.  gt4mi_comm wraps an RCCL communicator created from a 128-byte unique id
 * (gt4mi_comm_unique_id on one rank, distributed by the host program, e.g. torch.distributed).
 * A gt4mi_halo_plan holds, for one field shape, the boxes to send/receive in the (up to) two phases of the
 * exchange and owns the dense device staging buffers.  Two message tables are in use (the host builds them):
 *   two-phase     phase 0: I faces, phase 1: J faces including the I-halo columns (corners for free, 4 neighbours,
 *                 two dependent rounds of pack -> send/recv -> unpack);
 *   single-phase  everything in phase 0: 4 faces + 4 corner boxes to up to 8 neighbours, ONE round -- half the
 *                 latency; on a fully connected xGMI node the diagonal neighbours have links of their own.
 * Within a phase the k-th send to a peer pairs with the k-th receive that peer posts from this rank (RCCL
 * point-to-point ordering); at most 8 boxes per phase and direction. */
typedef struct gt4mi_comm gt4mi_comm;
typedef struct gt4mi_halo_plan gt4mi_halo_plan;

typedef struct gt4mi_halo_msg {
    int32_t peer;      /* rank of the neighbour                                                  */
    int32_t phase;     /* 0 or 1                                                                  */
    int64_t lo[3];     /* box start, in indices of the field array (NOT relative to the origin)  */
    int64_t extent[3]; /* box size                                                                */
} gt4mi_halo_msg;

int gt4mi_comm_unique_id(void* id128);
int gt4mi_comm_create(const void* id128, int nranks, int rank, gt4mi_comm** comm);
/* A communicator WITHOUT RCCL behind it (ranks that RCCL cannot join: two processes on one device; or no librccl at all):
 * plans created on it move their messages through the direct transport only (gt4mi_halo_plan_direct_*). */
int gt4mi_comm_create_local(int nranks, int rank, gt4mi_comm** comm);
int gt4mi_comm_destroy(gt4mi_comm* comm);
/* What RCCL reports for the communicator (ncclCommCount, ncclCommUserRank, ncclCommCuDevice); any pointer may be NULL.
 * Lets a benchmark line state how many ranks RCCL really joined. */
int gt4mi_comm_info(gt4mi_comm* comm, int* nranks, int* rank, int* device);
int gt4mi_halo_plan_create(gt4mi_comm* comm, int elem_size, const gt4mi_halo_msg* sends, int nsends,
                           const gt4mi_halo_msg* recvs, int nrecvs, gt4mi_halo_plan** plan);
int gt4mi_halo_plan_destroy(gt4mi_halo_plan* plan);
/* How the fused distributed steps (gt4mi_dist_*) built on this plan are scheduled; value -1 = the entry point's default.
 *   GT4MI_PLAN_SCHEDULE             GT4MI_SCHEDULE_JOIN:  main: pack, interior, [join], ring;  side: send/recv, unpack
 *                                   GT4MI_SCHEDULE_CHAIN: main: interior only;  side: pack, send/recv, unpack, ring -- no
 *                                   cross-stream wait on the critical path as long as the chain fits under the interior
 *                                   GT4MI_SCHEDULE_SWAP (gt4mi_dist_lap5_f64 and gt4mi_dist_hdiff_*): main: pack, send/recv, unpack,
 *                                   ring back to back;  side: the interior kernel; the caller's stream joins the interior at
 *                                   the end -- for shares so small that the chain, not the interior, is the critical path
 *                                   GT4MI_SCHEDULE_SWAP_PACKED: the same, the interior kernel forking off AFTER the pack (the
 *                                   pack of strided I faces runs alone, the send/recv kernel starts ahead of the interior)
 *                                   GT4MI_SCHEDULE_INLINE: everything on the caller's stream, no event: pack, interior, the rest
 *                                   of the exchange, ring -- made for the direct transport, whose pack kernel IS the transfer:
 *                                   the faces travel while the interior kernel runs and the unpack finds them there
 *   GT4MI_PLAN_INTERIOR_WG_PER_CU   at most this many workgroups of the INTERIOR kernel per CU while the exchange runs
 *                                   next to it (0 = no limit): an HBM-saturating kernel at full occupancy keeps tens of MB
 *                                   in flight and the send/recv kernel beside it waits ~10 us per memory access
 *   GT4MI_PLAN_DEFER_JOIN           1 (chain schedule only): gt4mi_dist_hdiff_* / gt4mi_dist_lap5_f64 return WITHOUT making
 *                                   the caller's stream wait for the side stream's chain; the caller joins with
 *                                   gt4mi_halo_exchange_end before anything consumes the result.  For INDEPENDENT applies:
 *                                   the interior of the next apply runs next to the exchange and ring of this one.
 *   GT4MI_PLAN_EDGE_COLUMNS         gt4mi_dist_hdiff_* / gt4mi_dist_lap5_f64: width of the W / E boxes left to the ring kernel
 *                                   (even; default 16, 8 for local domains narrower than 256 columns; the Laplacian's at most
 *                                   16): what the ring computes is taken off the interior kernel, a box of whole cache lines
 *                                   costs the memory system less than the 1 - 2 columns the stencil's reach requires, and the
 *                                   interior kernel keeps its 16-byte alignment.  gt4mi_dist_lap5_*: effective ONLY where
 *                                   gt4mi_dist_lap5_query reports edge_units = 0 -- where the edge units of csrc/lap5_edge.hip.h
 *                                   run (one receiving round, I-contiguous 16-byte aligned rows) the W / E boxes are exactly one
 *                                   16-byte lane wide on every schedule and transport, whatever this option says
 * Which combination is fastest depends on the links; bench.py measures them (config.calibration_ms_per_apply).
 *   GT4MI_PLAN_DIRECT_TIMEOUT_MS    direct transport: how long a device-side wait for a neighbour may take (milliseconds; 0 = the
 *                                   default: GT4MI_DIRECT_TIMEOUT_MS of the environment, else 30 000) before the plan FAILS, see below
 *   GT4MI_PLAN_DIRECT_FENCED        direct transport, 0 (default) / 1: FENCED MODE.  By default a pushed face is ordered before its
 *                                   flag by write-through stores + their acknowledgement, and the receiver's loads behind its flag
 *                                   load by issue order and cache-bypassing loads -- no fence, because a system-scope release next
 *                                   to an HBM-saturating interior kernel is expensive.  With 1 the pushing side
 *                                   does a system-scope RELEASE fence before it raises the flag and the receiving side a system-scope
 *                                   ACQUIRE fence behind its flag load: the ISA's own message-passing recipe, for links on which
 *                                   the default's assumptions have not been verified.  Both sides of a message must agree only
 *                                   in that each may be fenced or not independently (the modes interoperate); bench.py steps
 *                                   direct -> direct-fenced -> RCCL when its epoch-stamped self-check fails. */
enum { GT4MI_PLAN_SCHEDULE = 0, GT4MI_PLAN_INTERIOR_WG_PER_CU = 1, GT4MI_PLAN_DEFER_JOIN = 2, GT4MI_PLAN_EDGE_COLUMNS = 3,
       GT4MI_PLAN_TRANSPORT = 4, GT4MI_PLAN_DIRECT_TIMEOUT_MS = 5, GT4MI_PLAN_DIRECT_FENCED = 6 };
enum { GT4MI_TRANSPORT_RCCL = 0, GT4MI_TRANSPORT_DIRECT = 1 };
enum { GT4MI_SCHEDULE_JOIN = 0, GT4MI_SCHEDULE_CHAIN = 1, GT4MI_SCHEDULE_SWAP = 2, GT4MI_SCHEDULE_SWAP_PACKED = 3,
       GT4MI_SCHEDULE_INLINE = 4 };
int gt4mi_halo_plan_set_option(gt4mi_halo_plan* plan, int option, int value);
/* The DIRECT transport (GT4MI_PLAN_TRANSPORT = GT4MI_TRANSPORT_DIRECT; csrc/direct.hip.h): the pack kernel stores every face
 * straight into the neighbour's receive buffer (mapped with hipIpcOpenMemHandle; the neighbour may be this rank itself, another
 * process on this device, or another device of the node) and raises a flag there; the neighbour's unpack kernel waits for its
 * flags.  No send/recv kernel, two launches per phase.  Set-up, once per plan, by the host side that knows who the peers are:
 *   1. _direct_prepare on every rank: moves the plan's receive buffers into one exportable pool of fine-grained device memory
 *      (its first page holds the flag words) and fills `info` -- plain bytes to hand to the peers over any channel;
 *   2. _direct_layout: where receive (phase, index) sits in this rank's pool and which flag belongs to a message -- for the peers;
 *   3. _direct_connect for every message: sends[phase][index] lands at `peer_pool_offset` of the peer's pool and raises the peer's
 *      flag `peer_flag_index` (is_send = 1); after unpacking recvs[phase][index] this rank raises the SENDER's flag
 *      `peer_flag_index` (is_send = 0).  The k-th send to a peer pairs with the k-th receive that peer posted for this rank (RCCL's
 *      matching rule); `peer` = NULL: this rank itself;
 *   4. gt4mi_halo_plan_set_option(plan, GT4MI_PLAN_TRANSPORT, GT4MI_TRANSPORT_DIRECT) after every rank has connected.
 * FAILURE IS HARD: a device-side wait that runs out of time (GT4MI_PLAN_DIRECT_TIMEOUT_MS) copies and signals nothing and sets
 * the plan's error word (host memory the device writes); the NEXT call on the plan that touches the exchange -- gt4mi_halo_exchange*,
 * gt4mi_dist_*, gt4mi_halo_exchange_end -- reads it without synchronising and returns GT4MI_ERR_TIMEOUT, and so does every call
 * after it.  The call that enqueued the failing exchange has returned GT4MI_OK long before (everything is asynchronous): check
 * the status of the call that CONSUMES the result (_end, the next step, or _direct_status) before trusting ghost cells.
 * _direct_status synchronises the device first: whether a wait of any exchange started so far ran out of time.
 * DESTROYING a prepared plan is collective in effect: the neighbours' kernels write into this plan's pool ("consumed" adds, the
 * next pushes); call gt4mi_halo_plan_destroy only after every rank has finished its last exchange ON THE DEVICE (each rank
 * synchronises, then the ranks meet once on the host's control channel; gt4py_amd/distributed/native.py close()). */
typedef struct gt4mi_direct_info {
    char pool_handle[64];  /* hipIpcMemHandle_t of the pool: a page of flag words, then the receive buffers */
    int64_t pool_bytes, flag_words;
    int32_t pid, device;
} gt4mi_direct_info;
int gt4mi_halo_plan_direct_prepare(gt4mi_halo_plan* plan, gt4mi_direct_info* info);
int gt4mi_halo_plan_direct_layout(gt4mi_halo_plan* plan, int phase, int is_send, int index, int64_t* pool_offset, int* flag_index);
int gt4mi_halo_plan_direct_connect(gt4mi_halo_plan* plan, int phase, int is_send, int index, const gt4mi_direct_info* peer,
                                   int64_t peer_pool_offset, int peer_flag_index);
int gt4mi_halo_plan_direct_status(gt4mi_halo_plan* plan, int* timed_out, unsigned* exchanges);
/* 1 = the plan's side stream was verified to run concurrently with the caller's stream, 0 = no
 * concurrent stream could be found (the exchange still works, serialised), 2 = not probed yet.
 * HIP multiplexes streams onto a few hardware queues; the overlapped entry points probe on first use
 * and replace a side stream that shares the caller's queue. */
int gt4mi_halo_plan_concurrent(gt4mi_halo_plan* plan);
/* Enqueue the whole exchange of `field` on `stream` (stream-ordered, returns immediately). */
int gt4mi_halo_exchange(gt4mi_halo_plan* plan, const gt4mi_field* field, void* stream);
/* Overlapped form: _begin makes the plan's side stream wait for `main_stream`, enqueues the exchange
 * there and records completion; _end makes `main_stream` wait for that completion.  Work enqueued on
 * `main_stream` between the two calls (the interior kernel) runs concurrently with the exchange. */
int gt4mi_halo_exchange_begin(gt4mi_halo_plan* plan, const gt4mi_field* field, void* main_stream);
/* Optional: mark the fork point on `main_stream` NOW and enqueue the exchange later.  Lets the
 * caller enqueue the interior kernel before _begin, so the GPU is already busy while the host is
 * still issuing the pack / RCCL / unpack sequence; the next _begin then waits only for work that
 * was on `main_stream` before the fork. */
int gt4mi_halo_exchange_fork(gt4mi_halo_plan* plan, void* main_stream);
int gt4mi_halo_exchange_end(gt4mi_halo_plan* plan, void* main_stream);
/* One distributed apply of a 5-point stencil in a single call: exchange of `inp`'s halo (width 1)
 * overlapped with the interior kernel, then the boundary strips.  `sides` = bit mask of the sides
 * that have a neighbour: 1 = low I (W), 2 = high I (E), 4 = low J (S), 8 = high J (N).  Schedule (GT4MI_PLAN_SCHEDULE):
 * default GT4MI_SCHEDULE_SWAP (GT4MI_SCHEDULE_INLINE on the direct transport); whatever the schedule, work enqueued on `main_stream` after the call sees the whole result
 * (unless GT4MI_PLAN_DEFER_JOIN says otherwise).  Where the plan receives in ONE round (a single-phase table, or a grid cut along one
 * axis) and the fields are I-contiguous with 16-byte aligned rows, the unpack and the boundary strips are one kernel of units that read
 * the receive buffers themselves, and on the direct transport's GT4MI_SCHEDULE_INLINE the whole apply is ONE launch (csrc/lap5_edge.hip.h).
 * After the call `inp` has its ghost cells, as after gt4mi_halo_exchange. */
int gt4mi_dist_lap5_f64(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                        const gt4mi_field* out, int variant, int sides, void* main_stream);
/* Which launches gt4mi_dist_lap5_* would make for these arguments (nothing is enqueued): *edge_units = 1 when the unpack and the
 * boundary strips are the edge units of csrc/lap5_edge.hip.h (and, on the direct transport's GT4MI_SCHEDULE_INLINE, the whole apply
 * ONE launch), 0 when they are separate unpack and ring launches (two receiving rounds, a width that is no multiple of the 16-byte
 * lane, a layout without unit I stride ...).  The plan's item size selects float64 / float32. */
int gt4mi_dist_lap5_query(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int sides,
                          int* edge_units);
/* The same for float32 fields (a plan created for 4-byte items); `flags` as for gt4mi_lap5_f32 (GT4MI_LAP_LITERAL_F32). */
int gt4mi_dist_lap5_f32(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                        const gt4mi_field* out, int variant, int flags, int sides, void* main_stream);

/* One distributed apply of horizontal diffusion (the stencil of gt4mi_hdiff_*; BASELINE configs[4]) in a single call:
 *   main stream: pack of in_field's faces -> interior kernel (the domain minus a ring 2 points deep on every side
 *                that has a neighbour) ................................. join -> ring kernel (one launch, four boxes)
 *   side stream:                             RCCL send/recv -> unpack (-> second phase of a two-phase plan)
 * The plan must exchange faces 2 deep and include the corner cells (either message table above).  `sides` as above. */
int gt4mi_dist_hdiff_f64(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* in_field,
                         const gt4mi_field* out_field, const gt4mi_field* coeff, double coeff_scalar, int flags, int sides,
                         void* main_stream);
int gt4mi_dist_hdiff_f32(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* in_field,
                         const gt4mi_field* out_field, const gt4mi_field* coeff, double coeff_scalar, int flags, int sides,
                         void* main_stream);

/* Time-stepping form (out of step n is inp of step n+1): each call
 *   1. joins the exchange that delivered `inp`'s ghost cells (started by the previous call, or once by
 *      gt4mi_halo_exchange_begin(plan, first_input, ...) before the first step),
 *   2. computes the boundary strips of `out`,
 *   3. starts the exchange of `out`'s ghost cells on the plan's side stream,
 *   4. computes the interior of `out` concurrently with that exchange.
 * Nothing on the main stream ever waits for an exchange that has not had a whole interior kernel to
 * complete.  After the last step `out`'s ghost cells are (being) refreshed; gt4mi_halo_exchange_end
 * joins. */
int gt4mi_dist_lap5_f64_pipelined(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                                  const gt4mi_field* out, int variant, int sides, void* main_stream);
/* Communication-avoiding generalisation: the fields carry ghost regions `halo` >= 1 cells deep (the
 * plan exchanges faces that deep) and ONE exchange serves `halo` consecutive steps.  Call with
 * phase = 0, 1, ..., halo-1, 0, 1, ... :
 *   phase 0            joins the exchange that delivered `inp`'s ghost cells;
 *   phase < halo-1     one launch over the compute domain grown by (halo-1-phase) cells towards every
 *                      neighbour -- the ghost results it writes are valid inputs of the next step, no
 *                      communication at all;
 *   phase == halo-1    the pipelined step above: boundary strips (halo deep) of `out`, pack, then the
 *                      exchange of `out`'s ghost cells next to the interior kernel.
 * Per-step overhead of the exchange choreography is divided by `halo` for (halo-1)/2 redundant rows
 * per side on average.  halo = 1 is gt4mi_dist_lap5_f64_pipelined. */
int gt4mi_dist_lap5_f64_wide(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* inp,
                             const gt4mi_field* out, int variant, int sides, int halo, int phase,
                             void* main_stream);

/* Time-skewed form of the same: ONE call runs a whole cycle of `halo` steps, field_a -> field_b -> field_a ... (the
 * result is in field_b for odd `halo`, in field_a for even), boundary first:
 *   1. joins the exchange that delivered field_a's ghost cells (previous cycle, or gt4mi_halo_exchange_begin once);
 *   2. for step s = 1 .. halo: the band from halo - s points outside the domain to 2 halo - s points inside it
 *      (one ring launch each) -- after the last band the halo-deep faces of the result are final;
 *   3. packs them and starts their exchange on the side stream;
 *   4. for step s = 1 .. halo: the interior (the domain shrunk by 2 halo - s), `halo` kernels that all run next to the
 *      exchange.
 * Same redundant rows as the _wide form, but the exchange has `halo` interior kernels to hide behind instead of one.
 * Two buffers suffice: band s overwrites only what interior s - 1 no longer reads. */
int gt4mi_dist_lap5_f64_skewed(gt4mi_halo_plan* plan, const int64_t domain[3], const gt4mi_field* field_a,
                               const gt4mi_field* field_b, int variant, int sides, int halo, void* main_stream);

/* ---- run-time compiled stencils (generic executor) --------------------------------------------
 * Replaces the reference's per-stencil JIT build: setuptools + nvcc building a pybind11 extension
 * (/root/reference/src/gt4py/cartesian/backend/pyext_builder.py:176-303, driven by
 * backend/gtc_common.py:226-275) and that extension's `run_computation`
 * (backend/gtc_common.py:65-103).  The host generates HIP source for stencils outside the hand-written
 * families (gt4py_amd/cartesian/backend/hip_codegen.py); these entries compile it in-process with
 * hiprtc for gfx950 (always with -O3 -std=c++17 -ffp-contract=off; `options` are appended), load the
 * code object and launch kernels from it.  Compilation needs no GPU; load/launch do.
 *
 * gt4mi_rtc_compile: *code is malloc'ed by the library, release it with gt4mi_rtc_free.  `log`
 * (optional) receives the compiler log, truncated to log_size.
 * gt4mi_launch: `args` is the kernel-argument block (the generated kernel takes ONE struct by value;
 * the host lays it out with C rules), copied at launch.  grid is in workgroups. */
typedef struct gt4mi_module gt4mi_module;
int gt4mi_rtc_compile(const char* source, const char* name, const char* const* options, int n_options,
                      void** code, size_t* code_size, char* log, size_t log_size);
int gt4mi_rtc_free(void* code);
int gt4mi_module_load(const void* code, gt4mi_module** module);
int gt4mi_module_unload(gt4mi_module* module);
int gt4mi_module_function(gt4mi_module* module, const char* name, void** function);
/* What the compiler made of a loaded kernel: registers per lane, bytes of scratch (spills) per lane, static LDS bytes
 * per workgroup.  The host uses it to refuse kernel variants that spill.  Any pointer may be NULL. */
int gt4mi_function_info(void* function, int* registers, int* scratch_bytes, int* lds_bytes);
int gt4mi_launch(void* function, const uint32_t grid[3], const uint32_t block[3], const void* args,
                 size_t args_size, void* stream, gt4mi_exec_info* info);
/* The launches of one stencil call in one crossing of the boundary, in order, on one stream: n kernels,
 * grids / blocks as n consecutive triples, one argument block (of the same size) per launch.  A stencil of
 * several stages -- or one whose sequential block runs plane by plane, two launches per K level -- otherwise
 * pays the host language's call overhead per kernel.  Stops at the first failing launch. */
int gt4mi_launch_batch(int n, void* const* functions, const uint32_t* grids, const uint32_t* blocks,
                       const void* const* args, size_t args_size, void* stream, gt4mi_exec_info* info);

/* ---- measurement helper ---------------------------------------------------------------------
 * Streaming device copy of nbytes (multiple of 16) with 16-byte lanes: the "achievable HBM"
 * yardstick printed next to the stencil numbers. */
int gt4mi_stream_copy(const void* src, void* dst, size_t nbytes, void* stream);

/* ---- memory groups (ABI 7; new: the reference allocates through cupy and knows nothing of the device's memory system) -------------
 * MI355X's memory is not one uniformly interleaved pool: two big allocations either share a group of memory channels or they do
 * not, and nothing in the HIP API says which.  Kernels feel it -- two 1.3 GB fields written side by side: 5.0-6.4 TB/s in one group,
 * 6.8-7.0 TB/s in two; the fp64 Laplacian 512^3 +2.3 % with `in` and `out` in different groups; the tridiagonal solve 0.70 of the
 * HBM peak with its five fields dealt over two groups, 0.61 with all five in one (profiles/r5_memory_groups.txt).  This probe
 * measures it: ONE kernel writes both buffers the way a column kernel does; *gbs = bytes written to both per second / 1e9.
 * `b` may be NULL (one buffer alone).  OVERWRITES the first `bytes` (rounded down to planes of 8 MiB, at least 192 MiB) of both
 * buffers; synchronous on `stream`.  gt4py_amd/storage/placement.py uses it to deal big fields over the groups. */
int gt4mi_memory_write_probe(void* a, void* b, size_t bytes, int iterations, void* stream, double* gbs);

#ifdef __cplusplus
}
#endif
#endif /* GT4PY_AMD_H */
