// libgt4py_amd.so -- extern "C" entry points declared in include/gt4py_amd.h.
//
// Each stencil entry replaces the pybind11 `run_computation` the reference generates per stencil
// (/root/reference/src/gt4py/cartesian/backend/gtc_common.py:65-103): validate the fields against
// the domain, shift pointers by the origins (gtc_common.py:48 `sid::shift_sid_origin`), launch the
// hand-written gfx950 kernel on the caller's stream.
#include <chrono>
#include <cstring>

#include "common.hip.h"
#include "halo.hip.h"
#include "hdiff.hip.h"
#include "lap5.hip.h"
#include "tridiag.hip.h"

namespace {

inline double now_seconds() {
    using clock = std::chrono::steady_clock;
    return std::chrono::duration<double>(clock::now().time_since_epoch()).count();
}

struct Timer {
    gt4mi_exec_info* info;
    explicit Timer(gt4mi_exec_info* i) : info(i) {
        if (info) info->run_cpp_start_time = now_seconds();
    }
    ~Timer() {
        if (info) info->run_cpp_end_time = now_seconds();
    }
};

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
    Timer t(info);
    return gt4mi::lap5_run<double, double>(domain, inp, out, variant, static_cast<hipStream_t>(stream));
}

int gt4mi_lap5_f32(const int64_t domain[3], const gt4mi_field* inp, const gt4mi_field* out, int variant,
                   int flags, void* stream, gt4mi_exec_info* info) {
    Timer t(info);
    if (flags & GT4MI_LAP_LITERAL_F32)
        return gt4mi::lap5_run<float, float>(domain, inp, out, variant, static_cast<hipStream_t>(stream));
    return gt4mi::lap5_run<float, double>(domain, inp, out, variant, static_cast<hipStream_t>(stream));
}

int gt4mi_hdiff_f64(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                    const gt4mi_field* coeff, double coeff_scalar, int flags, void* stream,
                    gt4mi_exec_info* info) {
    Timer t(info);
    return gt4mi::hdiff_run<double>(domain, in_field, out_field, coeff, coeff_scalar, flags,
                                    static_cast<hipStream_t>(stream));
}

int gt4mi_hdiff_f32(const int64_t domain[3], const gt4mi_field* in_field, const gt4mi_field* out_field,
                    const gt4mi_field* coeff, double coeff_scalar, int flags, void* stream,
                    gt4mi_exec_info* info) {
    Timer t(info);
    return gt4mi::hdiff_run<float>(domain, in_field, out_field, coeff, coeff_scalar, flags,
                                   static_cast<hipStream_t>(stream));
}

int gt4mi_tridiag_f64(const int64_t domain[3], const gt4mi_field* inf, const gt4mi_field* diag,
                      const gt4mi_field* sup, const gt4mi_field* rhs, const gt4mi_field* out, void* stream,
                      gt4mi_exec_info* info) {
    Timer t(info);
    return gt4mi::tridiag_run<double>(domain, inf, diag, sup, rhs, out, static_cast<hipStream_t>(stream));
}

int gt4mi_tridiag_f32(const int64_t domain[3], const gt4mi_field* inf, const gt4mi_field* diag,
                      const gt4mi_field* sup, const gt4mi_field* rhs, const gt4mi_field* out, void* stream,
                      gt4mi_exec_info* info) {
    Timer t(info);
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

int gt4mi_stream_copy(const void* src, void* dst, size_t nbytes, void* stream) {
    if (src == nullptr || dst == nullptr) return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "stream_copy: null pointer");
    if (nbytes % 16 != 0 || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 != 0)
        return gt4mi::fail(GT4MI_ERR_INVALID_ARGUMENT, "stream_copy: pointers and size must be multiples of 16 bytes");
    const size_t nvec = nbytes / 16;
    if (nvec == 0) return GT4MI_OK;
    constexpr int UNROLL = 4;
    size_t blocks = (nvec + 256 * UNROLL - 1) / (256 * UNROLL);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((gt4mi::stream_copy_kernel<UNROLL, true>), dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const gt4mi::u32x4*>(src),
                       static_cast<gt4mi::u32x4*>(dst), nvec);
    GT4MI_HIP_CHECK(hipGetLastError());
    return GT4MI_OK;
}

}  // extern "C"
