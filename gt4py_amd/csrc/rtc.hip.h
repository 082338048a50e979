// Run-time compilation and launch of generated stencil kernels (the generic executor's native half).
//
// The reference JIT-builds one pybind11 extension per stencil with setuptools + nvcc
// (/root/reference/src/gt4py/cartesian/backend/pyext_builder.py:176-303, called from
// backend/gtc_common.py:226-275) and imports it.  Here the generated HIP source is compiled in-process
// with hiprtc into a gfx950 code object, loaded with hipModuleLoadData and launched with an opaque
// kernel-argument block.  libhiprtc is dlopen'ed on first use so that the hand-written-kernel path
// never pays for mapping the compiler.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.hip.h"

struct gt4mi_module {
    hipModule_t module = nullptr;
};

namespace gt4mi {

struct RtcApi {
    // hiprtcResult is an enum (int); hiprtcProgram is an opaque pointer
    using Program = void*;
    int (*CreateProgram)(Program*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*CompileProgram)(Program, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(Program, size_t*) = nullptr;
    int (*GetProgramLog)(Program, char*) = nullptr;
    int (*GetCodeSize)(Program, size_t*) = nullptr;
    int (*GetCode)(Program, char*) = nullptr;
    int (*DestroyProgram)(Program*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
    std::string error;

    bool load() {
        if (handle) return true;
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) {
            error = std::string("cannot dlopen libhiprtc.so: ") + dlerror();
            return false;
        }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(handle, n);
            if (!p) error = std::string("libhiprtc.so lacks ") + n;
            return p;
        };
        CreateProgram = reinterpret_cast<decltype(CreateProgram)>(sym("hiprtcCreateProgram"));
        CompileProgram = reinterpret_cast<decltype(CompileProgram)>(sym("hiprtcCompileProgram"));
        GetProgramLogSize = reinterpret_cast<decltype(GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
        GetProgramLog = reinterpret_cast<decltype(GetProgramLog)>(sym("hiprtcGetProgramLog"));
        GetCodeSize = reinterpret_cast<decltype(GetCodeSize)>(sym("hiprtcGetCodeSize"));
        GetCode = reinterpret_cast<decltype(GetCode)>(sym("hiprtcGetCode"));
        DestroyProgram = reinterpret_cast<decltype(DestroyProgram)>(sym("hiprtcDestroyProgram"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("hiprtcGetErrorString"));
        const bool ok = CreateProgram && CompileProgram && GetProgramLogSize && GetProgramLog && GetCodeSize &&
                        GetCode && DestroyProgram && GetErrorString;
        if (!ok) {
            dlclose(handle);
            handle = nullptr;
        }
        return ok;
    }
};

inline RtcApi& rtc_api() {
    static RtcApi api;
    return api;
}

inline std::mutex& rtc_mutex() {
    static std::mutex m;
    return m;
}

// Compile `source` for gfx950.  On success *code is malloc'ed (caller frees with free()).
inline int rtc_compile(const char* source, const char* name, const char* const* options, int n_options,
                       void** code, size_t* code_size, char* log, size_t log_size) {
    if (!source || !code || !code_size) return fail(GT4MI_ERR_INVALID_ARGUMENT, "rtc_compile: null argument");
    if (log && log_size) log[0] = '\0';
    std::lock_guard<std::mutex> guard(rtc_mutex());
    RtcApi& api = rtc_api();
    if (!api.load()) return fail(GT4MI_ERR_UNSUPPORTED, "rtc_compile: %s", api.error.c_str());
    RtcApi::Program prog = nullptr;
    int rc = api.CreateProgram(&prog, source, name ? name : "gt4mi_stencil.hip", 0, nullptr, nullptr);
    if (rc != 0) return fail(GT4MI_ERR_HIP, "hiprtcCreateProgram: %s", api.GetErrorString(rc));
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
    for (int i = 0; i < n_options; ++i) opts.push_back(options[i]);
    rc = api.CompileProgram(prog, (int)opts.size(), opts.data());
    size_t n = 0;
    if (log && log_size && api.GetProgramLogSize(prog, &n) == 0 && n > 1) {
        std::string full(n, '\0');
        api.GetProgramLog(prog, &full[0]);
        strncpy(log, full.c_str(), log_size - 1);
        log[log_size - 1] = '\0';
    }
    if (rc != 0) {
        api.DestroyProgram(&prog);
        return fail(GT4MI_ERR_HIP, "hiprtcCompileProgram: %s (see the log)", api.GetErrorString(rc));
    }
    size_t size = 0;
    rc = api.GetCodeSize(prog, &size);
    if (rc != 0 || size == 0) {
        api.DestroyProgram(&prog);
        return fail(GT4MI_ERR_HIP, "hiprtcGetCodeSize: %s", api.GetErrorString(rc));
    }
    char* buf = static_cast<char*>(malloc(size));
    if (!buf) {
        api.DestroyProgram(&prog);
        return fail(GT4MI_ERR_HIP, "rtc_compile: out of host memory");
    }
    rc = api.GetCode(prog, buf);
    api.DestroyProgram(&prog);
    if (rc != 0) {
        free(buf);
        return fail(GT4MI_ERR_HIP, "hiprtcGetCode: %s", api.GetErrorString(rc));
    }
    *code = buf;
    *code_size = size;
    return GT4MI_OK;
}

}  // namespace gt4mi
