// Shared device/host helpers for libx3dhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/x3dhip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void x3d_set_error(const char* fmt, ...);

#define X3D_CHECK_ARG(cond)                                                       \
    do {                                                                          \
        if (!(cond)) {                                                            \
            x3d_set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond); \
            return X3D_EINVAL;                                                    \
        }                                                                         \
    } while (0)

#define X3D_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            x3d_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return X3D_ELAUNCH;                                                   \
        }                                                                         \
    } while (0)

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15); every lane of the row gets the total.
// quad_perm / row_half_mirror / row_mirror butterflies: 4 VALU ops, no LDS.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// Sum over all 64 lanes of a wave (every lane gets the total).
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// 1 / (1 + e^-x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float act_fwd(float s, int act) {
    if (act == X3D_ACT_RELU) return s > 0.f ? s : 0.f;
    if (act == X3D_ACT_SWISH) return s * sigmoidf_(s);
    return s;
}

// derivative of the activation at pre-activation s
__device__ __forceinline__ float act_bwd(float s, int act) {
    if (act == X3D_ACT_RELU) return s > 0.f ? 1.f : 0.f;
    if (act == X3D_ACT_SWISH) {
        float sg = sigmoidf_(s);
        return sg * (1.f + s * (1.f - sg));   // x3d.py:83-84
    }
    return 1.f;
}

// Block-wide sum of `nval` per-thread values (nval <= 32) for 256-thread blocks.
// red must hold 4*nval floats.  Result valid in thread 0..nval-1 (value index = threadIdx.x).
template <int NVAL>
__device__ __forceinline__ void block_sum_256(float (&v)[NVAL], float* red, float (&out)[NVAL]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NVAL; ++i) {
        float s = wave_sum(v[i]);
        if (lane == 0) red[wave * NVAL + i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NVAL; ++i) out[i] = red[i] + red[NVAL + i] + red[2 * NVAL + i] + red[3 * NVAL + i];
    __syncthreads();
}
