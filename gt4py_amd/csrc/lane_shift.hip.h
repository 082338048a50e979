// Wave64 neighbour exchange in registers: lane l reads a value from lane l-1 or l+1.
//
// gfx950 (GFX9 family) still has the whole-wave DPP shifts `wave_shr:1` / `wave_shl:1`, one
// v_mov_b32_dpp per dword with no LDS traffic.  With bound_ctrl lanes shifted in from outside the wave read 0
// and the compiler needs no `v_mov_b32 dst, 0` in front of every shift (50 of 1434 vector instructions of the
// f32 horizontal-diffusion kernel; callers overwrite the edge lanes).  Verified on MI355X by `microbench dpp`.
#pragma once

#include <hip/hip_runtime.h>

namespace gt4mi {

__device__ __forceinline__ int lane_from_prev(int v) {  // lane l receives lane l-1
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}
__device__ __forceinline__ int lane_from_next(int v) {  // lane l receives lane l+1
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}

template <typename X, bool FROM_PREV>
__device__ __forceinline__ X lane_shift(X v) {
    if constexpr (sizeof(X) == 4) {
        int r = FROM_PREV ? lane_from_prev(__builtin_bit_cast(int, v))
                          : lane_from_next(__builtin_bit_cast(int, v));
        return __builtin_bit_cast(X, r);
    } else {
        static_assert(sizeof(X) == 8, "lane_shift: 4- or 8-byte types only");
        const long long b = __builtin_bit_cast(long long, v);
        int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
        lo = FROM_PREV ? lane_from_prev(lo) : lane_from_next(lo);
        hi = FROM_PREV ? lane_from_prev(hi) : lane_from_next(hi);
        const long long r = ((long long)hi << 32) | (unsigned int)lo;
        return __builtin_bit_cast(X, r);
    }
}

}  // namespace gt4mi
