// wave_reduce.hpp -- wave64 all-reduce helpers on DPP (no LDS traffic, unlike ds_bpermute shuffles).
#pragma once

#include <hip/hip_runtime.h>

namespace wavered {

// DPP controls: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- after the
// four steps every lane holds the reduction of its 16-lane row; the four rows are combined from
// readlane values.  All lanes end with the same result.
template <typename Op>
__device__ __forceinline__ float wave_reduce_f32(float x, Op op) {
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)));
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)));
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true)));
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true)));
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return op(op(r0, r1), op(r2, r3));
}

__device__ __forceinline__ int wave_min_i32(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true));
    return min(min(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)),
               min(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}

// the same four DPP steps alone: every lane ends with the reduction of its own 16-lane row
template <typename Op>
__device__ __forceinline__ float row_reduce_f32(float x, Op op) {
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)));
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)));
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true)));
    x = op(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true)));
    return x;
}

__device__ __forceinline__ int row_min_i32(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true));
    return x;
}

struct MaxOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return __builtin_fmaxf(a, b); }
};

}  // namespace wavered
