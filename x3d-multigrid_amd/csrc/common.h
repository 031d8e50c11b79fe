// Shared device/host helpers for libx3dhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/x3dhip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void x3d_set_error(const char* fmt, ...);
void x3d_note_kernel(const char* name);     // x3d_last_kernel(): which kernel template an entry point launched

// Options of the library (api.hip; set with x3d_set_option, include/x3dhip.h).  x3d_opt(id) is the value NOW: entry
// points and tile-count queries read it per call, never cache it.
enum X3DOpt {
    X3D_OPT_FB_GRID = 0,      // persistent grid of the fused stage 1-2 kernels (pwf.hip)                       512
    X3D_OPT_PW_PGRID,         // grid cap of the persistent tiled pointwise kernels (pw4 / pw5)                   512
    X3D_OPT_PW_NT4_MIN,       // streaming pointwise kernels: minimum workgroup count for the float4 form         256
    X3D_OPT_PW_NO_PERSIST,    // non-persistent LDS-tiled pw2 instead of pw4 / pw5                                 0
    X3D_OPT_DW_TH,            // channelwise kernels: maximum rows per tile                                        16
    X3D_OPT_DW_BALANCE,       // channelwise kernels: equal tile heights                                           1
    X3D_OPT_DW_NO_V2,         // channelwise kernels: element loads instead of float2 pairs on even widths         0
    X3D_OPT_NO_PW6,           // chunked pw4 instead of the whole-K forward pw6                                    0
    X3D_OPT_NO_PW7,           // chunked pw5 instead of the whole-K data gradient pw7                              0
    X3D_OPT_NO_PWFS,          // streaming pw3 instead of the stage 1-2 forward pw_fwd_stream                      0
    X3D_OPT_DGRAD_F32,        // exact fp32-MFMA data-gradient kernels                                             0
    X3D_OPT_WGRAD_F32,        // exact fp32-MFMA weight-gradient kernels                                           0
    X3D_OPT_BWD_TERMS,        // bf16 terms per fp32 operand in the backward GEMMs: 3 (fp32 level) or 2 (~2^-16)   3
    X3D_OPT_NO_WGRAD4,        // 128 x 64 weight-gradient tiles instead of the wide ones                           0
    X3D_OPT_WG_CPW,           // weight gradient: voxel chunks per workgroup                                       8
    X3D_OPT_WG_CAP,           // weight gradient: workgroup cap per conv                                           256
    X3D_OPT_STEM_WG_CAP,      // stem weight gradient: workgroups                                                  512
    X3D_OPT_DW_TSPLIT_WGS,    // channelwise kernels: launches of at most this many workgroups split T in two      256 (0: never)
    X3D_OPT_DW_CPB_MAX,       // channelwise kernels: maximum channels per workgroup                               16
    X3D_OPT_PW6_MIN_M,        // whole-K forward kernel pw6: smallest output-channel count it takes                96
    X3D_OPT_PW_TWO_TILES_K,   // whole-K kernels: padded K from which a wave takes two M tiles of one staged tile  320
    X3D_OPT_DW_TSPLIT_WGS_FWD,  // the same threshold for the forward channelwise kernel (14 x 14 planes: 16.9 -> 14.7 us)  512
    X3D_OPT_NO_PW8,           // round 4: non-persistent pw6 instead of the producer / consumer forward kernel pw8          0
    X3D_OPT_PW8_GRID,         // workgroups of the persistent producer / consumer kernels (0 = one per CU)           0
    X3D_OPT_PW8_MAX_K,        // largest padded K the persistent FORWARD kernel pw8 takes (0 = never; it wins at K <= 128)   128
    X3D_OPT_NO_SE_BWD_MERGE,  // separate reduce_tiles + se_bwd_sample launches instead of the merged per-sample kernel  0
    X3D_OPT_PW_WAVES16,       // whole-K kernels pw6 / pw7, 16-wave workgroups: 0 never, 1 K >= 320, 2 also > 16 M tiles, 3 also > 8   2
    X3D_OPT_DW_TQUAD_WGS,     // channelwise backward: launches whose TWO-segment form has at most this many workgroups use four  0
    X3D_OPT_DW_TQUAD_WGS_FWD, // the same for the forward kernel                                                           0
    X3D_OPT_COUNT
};
int x3d_opt(int id);
int x3d_cu_count();          // CUs of the current device (api.hip)

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

// Workgroup barrier that orders LDS only.  __syncthreads() compiles to "s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier": it also
// drains every global load a wave has in flight (the next chunk's prefetch) and every output store it has issued.  Where a
// workgroup never reads global memory it wrote itself, only the LDS needs ordering.
__device__ __forceinline__ void x3d_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// ---------------------------------------------------------------------------------------
// Activation storage: fp32, or bf16 for the wide (Cmid) tensors of a bottleneck in the mixed-storage mode (BASELINE
// config 5: bf16 storage / fp32 accumulate).  `bf` is a launch-uniform flag; `i` is the ELEMENT index.  Arithmetic is
// always fp32: bf16 -> fp32 is a 16-bit shift, fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32).
// ---------------------------------------------------------------------------------------
// Three-term bf16 split of a PAIR of fp32 values (the operands of the pointwise MFMA kernels):
// x = hi + mid + lo with hi = RN(x), mid = RN(x - hi), lo = RN(x - hi - mid); H / M / L hold (a, b) as packed bf16 (a in the low
// half).  Written so that the compiler emits ONE v_cvt_pk_bf16_f32 per term and pair, widens the halves back with a shift and
// a mask of the packed word, and subtracts with one v_pk_add_f32: 9 vector instructions per pair.  The plain
// (__bf16)x / (float)h cast form compiled to 12 (every low half converted a second time on its own: ISA of round 3's
// kernels) in kernels whose staging phase is bound by vector issue.  The values are those of the cast form, bit for bit.
typedef float x3d_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x3d_pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    const bf2 p = {(__bf16)a, (__bf16)b};
    unsigned u = __builtin_bit_cast(unsigned, p);
    asm("" : "+v"(u));            // opaque to the optimiser, which otherwise re-derives each half from a scalar conversion
    return u;
}
__device__ __forceinline__ void x3d_split3_pair(float a, float b, unsigned& H, unsigned& M, unsigned& L) {
    H = x3d_pack_bf16(a, b);
    const x3d_f2 r1 = x3d_f2{a, b} - x3d_f2{__uint_as_float(H << 16), __uint_as_float(H & 0xffff0000u)};
    M = x3d_pack_bf16(r1.x, r1.y);
    const x3d_f2 r2 = r1 - x3d_f2{__uint_as_float(M << 16), __uint_as_float(M & 0xffff0000u)};
    L = x3d_pack_bf16(r2.x, r2.y);
}
typedef __attribute__((ext_vector_type(4))) __bf16 x3d_bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 x3d_bf16x2;

__device__ __forceinline__ float4 ldx4(const void* p, size_t i, int bf) {
    if (bf) {
        const x3d_bf16x4 v = *reinterpret_cast<const x3d_bf16x4*>(reinterpret_cast<const __bf16*>(p) + i);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ float2 ldx2(const void* p, size_t i, int bf) {
    if (bf) {
        const x3d_bf16x2 v = *reinterpret_cast<const x3d_bf16x2*>(reinterpret_cast<const __bf16*>(p) + i);
        return make_float2((float)v[0], (float)v[1]);
    }
    return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ float ldx1(const void* p, size_t i, int bf) {
    if (bf) return (float)reinterpret_cast<const __bf16*>(p)[i];
    return reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void stx4(void* p, size_t i, int bf, float a, float b, float c, float d) {
    if (bf) {
        x3d_bf16x4 v;
        v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
        *reinterpret_cast<x3d_bf16x4*>(reinterpret_cast<__bf16*>(p) + i) = v;
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(a, b, c, d);
    }
}
__device__ __forceinline__ void stx2(void* p, size_t i, int bf, float a, float b) {
    if (bf) {
        x3d_bf16x2 v;
        v[0] = (__bf16)a; v[1] = (__bf16)b;
        *reinterpret_cast<x3d_bf16x2*>(reinterpret_cast<__bf16*>(p) + i) = v;
    } else {
        *reinterpret_cast<float2*>(reinterpret_cast<float*>(p) + i) = make_float2(a, b);
    }
}
__device__ __forceinline__ void stx1(void* p, size_t i, int bf, float a) {
    if (bf) reinterpret_cast<__bf16*>(p)[i] = (__bf16)a;
    else reinterpret_cast<float*>(p)[i] = a;
}

// Raw forms for kernels whose storage flag is a RUN-TIME value: the load and the widening are separate calls, so that all
// loads of a phase can be issued back to back.  (With the one-call forms above the bf16 -> fp32 conversion sits inside the
// flag's branch, right behind its load: the compiler then waits vmcnt(0) in that branch, once per load -- measured on the
// batched weight-gradient kernel: 41 vmcnt(0) instead of 5, +44 % time.)  A bf16 payload travels in the low components.
__device__ __forceinline__ float4 ldx4_raw(const void* p, size_t i, int bf) {
    if (bf) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(p) + i);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), 0.f, 0.f);
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ float4 ldo4_raw(const char* base, unsigned eoff, int bf) {
    if (bf) {
        const uint2 v = *reinterpret_cast<const uint2*>(base + eoff * 2u);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), 0.f, 0.f);
    }
    return *reinterpret_cast<const float4*>(base + eoff * 4u);
}
__device__ __forceinline__ float2 ldx2_raw(const void* p, size_t i, int bf) {
    if (bf) {
        const unsigned v = *reinterpret_cast<const unsigned*>(reinterpret_cast<const __bf16*>(p) + i);
        return make_float2(__uint_as_float(v), 0.f);
    }
    return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ float ldo1_raw(const char* base, unsigned eoff, int bf) {
    if (bf) return __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(base + eoff * 2u));
    return *reinterpret_cast<const float*>(base + eoff * 4u);
}
// branch-free widening (selects on a uniform flag)
__device__ __forceinline__ float4 widen4(float4 r, int bf) {
    const unsigned a = __float_as_uint(r.x), b = __float_as_uint(r.y);
    float4 w;
    w.x = bf ? __uint_as_float(a << 16) : r.x;
    w.y = bf ? __uint_as_float(a & 0xffff0000u) : r.y;
    w.z = bf ? __uint_as_float(b << 16) : r.z;
    w.w = bf ? __uint_as_float(b & 0xffff0000u) : r.w;
    return w;
}
__device__ __forceinline__ float2 widen2(float2 r, int bf) {
    const unsigned a = __float_as_uint(r.x);
    return make_float2(bf ? __uint_as_float(a << 16) : r.x, bf ? __uint_as_float(a & 0xffff0000u) : r.y);
}
__device__ __forceinline__ float widen1(float r, int bf) { return bf ? __uint_as_float(__float_as_uint(r) << 16) : r; }

// What a consumer reads back from a tensor stored with flag `bf`: statistics that describe a stored tensor (BN sums, BN
// backward sums) are taken from these values, so the BN that follows sees the statistics of the tensor it normalises.
__device__ __forceinline__ float stored(float v, int bf) { return bf ? (float)(__bf16)v : v; }

// The same with a wave-uniform base (SGPR pair, already advanced to the sample in BYTES: see mx_base) and a 32-bit
// element offset (one VGPR): `global_load ... v_off, s[base]`.
__device__ __forceinline__ const char* mx_base(const void* p, size_t elems, int bf) {
    return reinterpret_cast<const char*>(p) + elems * (bf ? 2u : 4u);
}
__device__ __forceinline__ char* mx_base(void* p, size_t elems, int bf) {
    return reinterpret_cast<char*>(p) + elems * (bf ? 2u : 4u);
}
__device__ __forceinline__ float4 ldo4(const char* base, unsigned eoff, int bf) {
    if (bf) {
        const x3d_bf16x4 v = *reinterpret_cast<const x3d_bf16x4*>(base + eoff * 2u);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    return *reinterpret_cast<const float4*>(base + eoff * 4u);
}
__device__ __forceinline__ float2 ldo2(const char* base, unsigned eoff, int bf) {
    if (bf) {
        const x3d_bf16x2 v = *reinterpret_cast<const x3d_bf16x2*>(base + eoff * 2u);
        return make_float2((float)v[0], (float)v[1]);
    }
    return *reinterpret_cast<const float2*>(base + eoff * 4u);
}
__device__ __forceinline__ float ldo1(const char* base, unsigned eoff, int bf) {
    if (bf) return (float)*reinterpret_cast<const __bf16*>(base + eoff * 2u);
    return *reinterpret_cast<const float*>(base + eoff * 4u);
}
__device__ __forceinline__ void sto4(char* base, unsigned eoff, int bf, float a, float b, float c, float d) {
    if (bf) {
        x3d_bf16x4 v;
        v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
        *reinterpret_cast<x3d_bf16x4*>(base + eoff * 2u) = v;
    } else {
        *reinterpret_cast<float4*>(base + eoff * 4u) = make_float4(a, b, c, d);
    }
}
__device__ __forceinline__ void sto2(char* base, unsigned eoff, int bf, float a, float b) {
    if (bf) {
        x3d_bf16x2 v;
        v[0] = (__bf16)a; v[1] = (__bf16)b;
        *reinterpret_cast<x3d_bf16x2*>(base + eoff * 2u) = v;
    } else {
        *reinterpret_cast<float2*>(base + eoff * 4u) = make_float2(a, b);
    }
}
__device__ __forceinline__ void sto1(char* base, unsigned eoff, int bf, float a) {
    if (bf) *reinterpret_cast<__bf16*>(base + eoff * 2u) = (__bf16)a;
    else *reinterpret_cast<float*>(base + eoff * 4u) = a;
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
