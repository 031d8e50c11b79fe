// GPU-side clip input pipeline: what the reference does per frame on CPU workers --
//   MultiScaleRandomCropMultigrid (crop + PIL bilinear resize)  transforms/spatial_transforms.py:480-495
//   RandomHorizontalFlip                                        :334-346
//   ToTensor(255) + Normalize(mean, std)                        :44-83,106-116
//   stack(frames).permute(1,0,2,3)                              kinetics_multigrid.py:249-253
// plus the frame selection of TemporalRandomCrop (indices computed on the host, integer logic) --
// as two kernels over decoded uint8 frames resident in HBM.
//
// Bit-exact with Pillow's 8-bit bilinear resample (libImaging/Resample.c): separable, horizontal pass
// first into a uint8 intermediate, 22-bit fixed-point coefficient tables (built on the host in double
// precision exactly as precompute_coeffs / normalize_coeffs_8bpc do), round-half-up, clip to [0, 255].
// Byte/integer work, HBM bound: one thread per output pixel (3 channels), coalesced along x.
#include "common.h"
#include <stdint.h>

namespace {

constexpr int CLIP_PRECISION_BITS = 32 - 8 - 2;

// must match X3DClipJob in include/x3dhip.h
struct ClipJob {
    const uint8_t* src;       // [Tsrc][Hs][Ws][3] decoded frames
    uint8_t* tmp;             // [T][crop][out][3] horizontal-pass intermediate (caller-provided scratch)
    float* dst;               // [3][T][out][out] (the sample's slice of the NCTHW batch)
    const int32_t* kk;        // [out][ksize] coefficients (same table for both passes: square crop, square output)
    const int32_t* bounds;    // [out][2]  (first input index, tap count)
    const int32_t* frames;    // [T] 0-based source frame of each output frame
    int Hs, Ws, x1, y1, crop, out, ksize, T, flip, pad;
};

__device__ __forceinline__ int clip8(int v) {
    v >>= CLIP_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[t][y][xx][c] = clip8(sum_i src[frame t][y1 + y][x1 + x0 + i][c] * kk[xx][i])
__global__ __launch_bounds__(256) void clip_hpass_kernel(const ClipJob* __restrict__ jobs) {
    const ClipJob J = jobs[blockIdx.z];
    const int t = blockIdx.y;
    if (t >= J.T) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= J.crop * J.out) return;
    const int y = idx / J.out, xx = idx - y * J.out;
    const int x0 = J.bounds[xx * 2], n = J.bounds[xx * 2 + 1];
    const int32_t* k = J.kk + (size_t)xx * J.ksize;
    const uint8_t* row = J.src + (((size_t)J.frames[t] * J.Hs + (J.y1 + y)) * J.Ws + (J.x1 + x0)) * 3;
    int a0 = 1 << (CLIP_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int i = 0; i < n; ++i) {
        const int kv = k[i];
        a0 += (int)row[i * 3] * kv;
        a1 += (int)row[i * 3 + 1] * kv;
        a2 += (int)row[i * 3 + 2] * kv;
    }
    uint8_t* o = J.tmp + (((size_t)t * J.crop + y) * J.out + xx) * 3;
    o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
}

// vertical pass + flip + ToTensor(255) + Normalize: dst[c][t][yy][xx'] = ((v / 255) - mean[c]) / std[c]
__global__ __launch_bounds__(256) void clip_vpass_kernel(const ClipJob* __restrict__ jobs, float m0, float m1, float m2,
                                                         float s0, float s1, float s2) {
    const ClipJob J = jobs[blockIdx.z];
    const int t = blockIdx.y;
    if (t >= J.T) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= J.out * J.out) return;
    const int yy = idx / J.out, xx = idx - yy * J.out;
    const int y0 = J.bounds[yy * 2], n = J.bounds[yy * 2 + 1];
    const int32_t* k = J.kk + (size_t)yy * J.ksize;
    const uint8_t* col = J.tmp + (((size_t)t * J.crop + y0) * J.out + xx) * 3;
    const size_t pitch = (size_t)J.out * 3;
    int a0 = 1 << (CLIP_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int i = 0; i < n; ++i) {
        const int kv = k[i];
        a0 += (int)col[i * pitch] * kv;
        a1 += (int)col[i * pitch + 1] * kv;
        a2 += (int)col[i * pitch + 2] * kv;
    }
    const int xo = J.flip ? J.out - 1 - xx : xx;
    const size_t plane = (size_t)J.out * J.out, cstride = (size_t)J.T * plane;
    float* d = J.dst + (size_t)t * plane + (size_t)yy * J.out + xo;
    // img.float().div(255) ; t.sub_(m).div_(s)  -- three separately rounded fp32 operations
    d[0] = ((float)clip8(a0) / 255.0f - m0) / s0;
    d[cstride] = ((float)clip8(a1) / 255.0f - m1) / s1;
    d[2 * cstride] = ((float)clip8(a2) / 255.0f - m2) / s2;
}

}  // namespace

extern "C" size_t x3d_clip_job_bytes(void) { return sizeof(ClipJob); }

extern "C" int x3d_clip_preprocess(const void* jobs, int njobs, int max_T, int max_crop, int max_out, const float* mean,
                                   const float* stdv, void* stream) {
    X3D_CHECK_ARG(jobs && mean && stdv);
    X3D_CHECK_ARG(njobs > 0 && njobs <= 65535 && max_T > 0 && max_T <= 65535 && max_crop > 0 && max_out > 0);
    X3D_CHECK_ARG((long long)max_crop * max_out < (1LL << 31));
    hipStream_t s = (hipStream_t)stream;
    const ClipJob* J = (const ClipJob*)jobs;
    hipLaunchKernelGGL(clip_hpass_kernel, dim3(cdiv(max_crop * max_out, 256), max_T, njobs), dim3(256), 0, s, J);
    hipLaunchKernelGGL(clip_vpass_kernel, dim3(cdiv(max_out * max_out, 256), max_T, njobs), dim3(256), 0, s, J, mean[0],
                       mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
