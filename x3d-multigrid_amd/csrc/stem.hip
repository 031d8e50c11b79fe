// Stem: conv1_s (dense 1x3x3, stride (1,2,2), 3 -> 24; x3d.py:196-201,317) and conv1_t
// (depthwise temporal 5x1x1, pad (2,0,0); x3d.py:202-208,318), forward and backward.
// No BN sits between the two convolutions, so conv1_s emits raw output only; conv1_t's
// epilogue carries the statistics for bn1 (x3d.py:209,319).
#include <cstdlib>
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------
// conv1_s forward: one thread per output voxel, all Cout channels (weights are wave-uniform
// -> scalar loads); the 27 input taps are loaded once and reused for every output channel.
// ---------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void stem133_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int Cout, int T, int H, int W,
                                                          int Ho, int Wo) {
    const int n = blockIdx.z, t = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int ho = p / Wo, wo = p - ho * Wo;
    float v[CIN * 9];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        const float* px = x + (((size_t)n * CIN + ci) * T + t) * (size_t)H * W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = 2 * ho - 1 + kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = 2 * wo - 1 + kw;
                v[ci * 9 + kh * 3 + kw] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? px[(size_t)hi * W + wi] : 0.f;
            }
        }
    }
    // four output channels per round: their 4 x 27 wave-uniform weights are fetched by one batch of scalar loads and the
    // four FMA chains are independent (a single chain waits for each scalar load batch and for its own FMA latency)
    int co = 0;
    for (; co + 3 < Cout; co += 4) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float* wc = w + (size_t)co * CIN * 9;
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) {
            s0 = fmaf(wc[k], v[k], s0);
            s1 = fmaf(wc[CIN * 9 + k], v[k], s1);
            s2 = fmaf(wc[2 * CIN * 9 + k], v[k], s2);
            s3 = fmaf(wc[3 * CIN * 9 + k], v[k], s3);
        }
        float* py = y + (((size_t)n * Cout + co) * T + t) * (size_t)Ho * Wo + p;
        const size_t cs = (size_t)T * Ho * Wo;
        py[0] = s0; py[cs] = s1; py[2 * cs] = s2; py[3 * cs] = s3;
    }
    for (; co < Cout; ++co) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) s = fmaf(w[co * CIN * 9 + k], v[k], s);
        y[(((size_t)n * Cout + co) * T + t) * (size_t)Ho * Wo + p] = s;
    }
}

// ---------------------------------------------------------------------------------------
// conv1_s backward-weight as an MFMA GEMM: dW[co][j] = sum_p dy[co][p] * patch[j][p],
// j = ci*9 + kh*3 + kw (27 -> 32), co (24 -> 32).  A tile is a run of 128 output voxels of ONE output row; each of the 4
// waves owns one 16x16 tile of dW.  dy is staged in LDS ([32][132]); the input is staged as the 3 x Cin input rows the
// run touches ([ci*3 + kh][264 columns], coalesced float4 row loads, zero outside the image) and the im2col operand is
// read straight from that image (column 2 p + kw of row (ci, kh)) -- round 1 gathered every tap from global memory,
// 16 scalar loads per thread and tile at stride 2: 273 MB fetched for 231 MB of tensors, 138 us for a 46 us problem.
// ---------------------------------------------------------------------------------------
constexpr int SW_PT = 128, SW_LD = 132;
constexpr int SX_LD = 264;                 // input columns 2 wo0 - 4 .. 2 wo0 + 259 of a run starting at output column wo0
constexpr int SX_ROWS = 10;                // 3 x Cin (<= 9) image rows + one row of zeros for the padded j >= 27

__global__ __launch_bounds__(256) void stem133_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ wpartial, int N, int Cin, int Cout,
                                                            int T, int H, int W, int Ho, int Wo, int groups) {
    __shared__ __attribute__((aligned(16))) float Ld[32 * SW_LD];
    __shared__ __attribute__((aligned(16))) float Xt[SX_ROWS * SX_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int wr = wave >> 1, wc = wave & 1;
    const int J = Cin * 9;
    const int HWo = Ho * Wo;
    const int chunks = cdiv(Wo, SW_PT);                    // runs per output row
    const int tiles_per_plane = Ho * chunks;
    const int total = N * T * tiles_per_plane;
    const bool vecx = (W & 3) == 0;                        // aligned float4 row loads
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < SX_LD; i += 256) Xt[(SX_ROWS - 1) * SX_LD + i] = 0.f;

    // Staging roles.  dy: thread owns voxel column pp = tid & 127 and the 16 rows r0, r0 + 2, ...; input image: float4
    // granule g = tid + 256 i of the 3 Cin x 66 granules.  The loads of the NEXT tile are in flight during the MFMAs.
    // Everything that does not depend on the tile is computed once (the first version of this kernel spent most of its
    // instructions on 64-bit index arithmetic and tile-decoding divisions: it ran at the same 100 us with 2, 3 or 4
    // workgroups per CU).  A workgroup walks a CONTIGUOUS range of tiles (plane-major, then output row, then run) with
    // incremental counters; neighbouring output rows share an input row, which then comes from cache.
    const int pp = tid & (SW_PT - 1), r0 = tid >> 7;
    constexpr int GPR = SX_LD / 4;                         // granules per image row
    float dv[16];
    float4 xr[3];
    unsigned droff[16];                                    // dy row offsets inside a sample (a sample's dy is < 2^32 elements)
#pragma unroll
    for (int i = 0; i < 16; ++i) droff[i] = (unsigned)min(2 * i + r0, Cout - 1) * (unsigned)T * (unsigned)HWo;
    unsigned xcoff[3];                                     // channel offset of the granule's image row inside a sample
    int xkh[3], xc4[3];
    bool xgv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int g = tid + 256 * i;
        const int row = min(g / GPR, Cin * 3 - 1);
        xc4[i] = (g - (g / GPR) * GPR) * 4;
        xkh[i] = row % 3;
        xcoff[i] = (unsigned)(row / 3) * (unsigned)T * (unsigned)H * (unsigned)W;
        xgv[i] = g < Cin * 3 * GPR;
    }
    const int t_begin = (int)(((long long)blockIdx.x * total) / groups), t_end = (int)(((long long)(blockIdx.x + 1) * total) / groups);
    // position of the tile being LOADED (one ahead of the tile being computed)
    int l_plane = t_begin / tiles_per_plane, l_ho, l_ch;
    {
        const int rem = t_begin - l_plane * tiles_per_plane;
        l_ho = rem / chunks;
        l_ch = rem - l_ho * chunks;
    }
    bool l_pv = false;                                     // voxel column of the loaded tile inside the output row
    auto load_tile = [&]() {
        const int ho = l_ho, wo0 = l_ch * SW_PT;
        const int n = l_plane / T, t = l_plane - n * T;
        const int wo = wo0 + pp;
        l_pv = wo < Wo;
        const float* dyb = dy + ((size_t)n * Cout * T + t) * (size_t)HWo + (size_t)ho * Wo + (l_pv ? wo : wo0);
#pragma unroll
        for (int i = 0; i < 16; ++i) dv[i] = dyb[droff[i]];
        const int wib = 2 * wo0 - 4;                       // first staged input column (multiple of 4)
        const float* xb = x + ((size_t)n * Cin * T + t) * (size_t)H * W;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hi = 2 * ho - 1 + xkh[i], wi = wib + xc4[i];
            const bool rok = xgv[i] && hi >= 0 && hi < H;
            const float* px = xb + xcoff[i] + (size_t)min(max(hi, 0), H - 1) * W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vecx) {
                const bool ok = rok && wi >= 0 && wi + 3 < W;            // W % 4 == 0 and wi % 4 == 0: all four columns or none
                const float4 t4 = *reinterpret_cast<const float4*>(px + (ok ? wi : 0));
                if (ok) v = t4;
            } else {
                const float e0 = px[min(max(wi, 0), W - 1)], e1 = px[min(max(wi + 1, 0), W - 1)];
                const float e2 = px[min(max(wi + 2, 0), W - 1)], e3 = px[min(max(wi + 3, 0), W - 1)];
                v.x = (rok && wi >= 0 && wi < W) ? e0 : 0.f;
                v.y = (rok && wi + 1 >= 0 && wi + 1 < W) ? e1 : 0.f;
                v.z = (rok && wi + 2 >= 0 && wi + 2 < W) ? e2 : 0.f;
                v.w = (rok && wi + 3 >= 0 && wi + 3 < W) ? e3 : 0.f;
            }
            xr[i] = v;
        }
        if (++l_ch == chunks) { l_ch = 0; if (++l_ho == Ho) { l_ho = 0; ++l_plane; } }
    };
    // B operand: lane (q, r) of wave (wr, wc) needs patch row j = wc * 16 + r at voxels kk * 16 + 4 q + e: column
    // 2 p + kw + 3 of image row (ci, kh) (staged column = input column - wib; input column = 2 (wo0 + p) - 1 + kw)
    const int jb = wc * 16 + r;
    const int jci = jb / 9, jk = jb - jci * 9, jkh = jk / 3, jkw = jk - jkh * 3;
    const float* xrow = Xt + (jb < J ? (jci * 3 + jkh) : (SX_ROWS - 1)) * SX_LD + (jb < J ? jkw + 3 : 0);

    if (t_begin < t_end) load_tile();
    for (int tl = t_begin; tl < t_end; ++tl) {
        __syncthreads();                       // previous tile's fragments read
        const bool pv = l_pv;                  // of the tile loaded last = the one staged now
#pragma unroll
        for (int i = 0; i < 16; ++i) Ld[(2 * i + r0) * SW_LD + pp] = (pv && 2 * i + r0 < Cout) ? dv[i] : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int g = tid + 256 * i;
            if (g < Cin * 3 * GPR) *reinterpret_cast<float4*>(&Xt[(g / GPR) * SX_LD + (g - (g / GPR) * GPR) * 4]) = xr[i];
        }
        __syncthreads();
        if (tl + 1 < t_end) load_tile();
#pragma unroll 2
        for (int kk = 0; kk < SW_PT / 16; ++kk) {
            const float4 av = *reinterpret_cast<const float4*>(&Ld[(wr * 16 + r) * SW_LD + kk * 16 + 4 * q]);
            const float* xb = xrow + 2 * (kk * 16 + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, xb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, xb[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, xb[4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, xb[6], acc, 0, 0, 0);
        }
    }
    float* out = wpartial + (size_t)blockIdx.x * Cout * J;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int co = wr * 16 + 4 * q + e, j = wc * 16 + r;
        if (co < Cout && j < J) out[(size_t)co * J + j] = acc[e];
    }
}

// ---------------------------------------------------------------------------------------
// conv1_t (depthwise 5x1x1).  Pure stream along H*W: each thread owns 4 consecutive voxels
// of a plane and marches along T with a 5-deep register window (every input read once).
// ---------------------------------------------------------------------------------------
constexpr int T5_TILE = 1024;

template <bool VEC>
__global__ __launch_bounds__(256) void dw5t_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int C, int T, int HW,
                                                       float* __restrict__ partial, int tiles) {
    __shared__ float red[4 * 2];
    // 1-D grid (rows x tiles, tile fastest): N * C is not bounded by the 65535 limit of grid.y
    const int row = blockIdx.x / tiles, tix = blockIdx.x - row * tiles;            // row = n*C + c
    const int c = row % C;
    const int p = tix * T5_TILE + threadIdx.x * 4;
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = w[c * 5 + k];
    const float* px = x + (size_t)row * T * HW;
    float* py = y + (size_t)row * T * HW;
    auto load = [&](int t) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T && p < HW) {
            const float* q = px + (size_t)t * HW + p;
            if (VEC) v = *reinterpret_cast<const float4*>(q);
            else {
                v.x = q[0];
                if (p + 1 < HW) v.y = q[1];
                if (p + 2 < HW) v.z = q[2];
                if (p + 3 < HW) v.w = q[3];
            }
        }
        return v;
    };
    float4 win[5];
    win[0] = win[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    win[2] = load(0); win[3] = load(1); win[4] = load(2);
    float s[2] = {0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        float4 o;
        o.x = fmaf(wk[0], win[0].x, fmaf(wk[1], win[1].x, fmaf(wk[2], win[2].x, fmaf(wk[3], win[3].x, wk[4] * win[4].x))));
        o.y = fmaf(wk[0], win[0].y, fmaf(wk[1], win[1].y, fmaf(wk[2], win[2].y, fmaf(wk[3], win[3].y, wk[4] * win[4].y))));
        o.z = fmaf(wk[0], win[0].z, fmaf(wk[1], win[1].z, fmaf(wk[2], win[2].z, fmaf(wk[3], win[3].z, wk[4] * win[4].z))));
        o.w = fmaf(wk[0], win[0].w, fmaf(wk[1], win[1].w, fmaf(wk[2], win[2].w, fmaf(wk[3], win[3].w, wk[4] * win[4].w))));
        if (p < HW) {
            float* q = py + (size_t)t * HW + p;
            if (VEC) *reinterpret_cast<float4*>(q) = o;
            else {
                q[0] = o.x;
                if (p + 1 < HW) q[1] = o.y; else o.y = 0.f;
                if (p + 2 < HW) q[2] = o.z; else o.z = 0.f;
                if (p + 3 < HW) q[3] = o.w; else o.w = 0.f;
            }
            s[0] += (o.x + o.y) + (o.z + o.w);
            s[1] = fmaf(o.x, o.x, fmaf(o.y, o.y, fmaf(o.z, o.z, fmaf(o.w, o.w, s[1]))));
        }
        win[0] = win[1]; win[1] = win[2]; win[2] = win[3]; win[3] = win[4];
        win[4] = load(t + 3);
    }
    if (partial != nullptr) {
        float o2[2];
        block_sum_256<2>(s, red, o2);
        if (threadIdx.x == 0) {
            partial[((size_t)row * tiles + tix) * 2] = o2[0];
            partial[((size_t)row * tiles + tix) * 2 + 1] = o2[1];
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void dw5t_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                       const float* __restrict__ cb, const float* __restrict__ w,
                                                       const float* __restrict__ x, float* __restrict__ dx,
                                                       float* __restrict__ wpartial, int C, int T, int HW, int tiles) {
    __shared__ float red[4 * 5];
    const int row = blockIdx.x / tiles, tix = blockIdx.x - row * tiles;
    const int c = row % C;
    const int p = tix * T5_TILE + threadIdx.x * 4;
    const float k0 = cb[(size_t)row * 3], k1 = cb[(size_t)row * 3 + 1], k2 = cb[(size_t)row * 3 + 2];
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = w[c * 5 + k];
    const size_t base = (size_t)row * T * HW;
    auto ld4 = [&](const float* src, int t) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T && p < HW) {
            const float* q = src + base + (size_t)t * HW + p;
            if (VEC) v = *reinterpret_cast<const float4*>(q);
            else {
                v.x = q[0];
                if (p + 1 < HW) v.y = q[1];
                if (p + 2 < HW) v.z = q[2];
                if (p + 3 < HW) v.w = q[3];
            }
        }
        return v;
    };
    auto dyv = [&](int t) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T && p < HW) {
            const float4 gv = ld4(g, t), av = ld4(a, t);
            v.x = fmaf(k0, gv.x, fmaf(k1, av.x, k2));
            if (p + 1 < HW) v.y = fmaf(k0, gv.y, fmaf(k1, av.y, k2));
            if (p + 2 < HW) v.z = fmaf(k0, gv.z, fmaf(k1, av.z, k2));
            if (p + 3 < HW) v.w = fmaf(k0, gv.w, fmaf(k1, av.w, k2));
        }
        return v;
    };
    // forward: y[t] = sum_k w[k] x[t+k-2]  =>  dx[t] = sum_k w[k] dY[t-k+2];  dW[k] = sum_t dY[t] x[t+k-2]
    // window d[j] = dY[t-2+j], j = 0..4  -> dx[t] = sum_k w[k] d[4-k]
    float4 d[5];
    d[0] = d[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    d[2] = dyv(0); d[3] = dyv(1); d[4] = dyv(2);
    float dw[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        float4 o;
        o.x = fmaf(wk[0], d[4].x, fmaf(wk[1], d[3].x, fmaf(wk[2], d[2].x, fmaf(wk[3], d[1].x, wk[4] * d[0].x))));
        o.y = fmaf(wk[0], d[4].y, fmaf(wk[1], d[3].y, fmaf(wk[2], d[2].y, fmaf(wk[3], d[1].y, wk[4] * d[0].y))));
        o.z = fmaf(wk[0], d[4].z, fmaf(wk[1], d[3].z, fmaf(wk[2], d[2].z, fmaf(wk[3], d[1].z, wk[4] * d[0].z))));
        o.w = fmaf(wk[0], d[4].w, fmaf(wk[1], d[3].w, fmaf(wk[2], d[2].w, fmaf(wk[3], d[1].w, wk[4] * d[0].w))));
        // x[t] pairs with dY[t - k + 2] for tap k:  dW[k] += x[t] * d[4-k]
        const float4 xv = ld4(x, t);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            dw[k] += fmaf(xv.x, d[4 - k].x, fmaf(xv.y, d[4 - k].y, fmaf(xv.z, d[4 - k].z, xv.w * d[4 - k].w)));
        if (p < HW) {
            float* q = dx + base + (size_t)t * HW + p;
            if (VEC) *reinterpret_cast<float4*>(q) = o;
            else {
                q[0] = o.x;
                if (p + 1 < HW) q[1] = o.y;
                if (p + 2 < HW) q[2] = o.z;
                if (p + 3 < HW) q[3] = o.w;
            }
        }
        d[0] = d[1]; d[1] = d[2]; d[2] = d[3]; d[3] = d[4];
        d[4] = dyv(t + 3);
    }
    float o5[5];
    block_sum_256<5>(dw, red, o5);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k)   // layout [N][tiles][C][5]: a plain group sum over (n, tile) finishes it
            wpartial[((((size_t)(row / C)) * tiles + tix) * C + c) * 5 + k] = o5[k];
    }
}

}  // namespace

extern "C" int x3d_stem133_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int T, int H,
                               int W, void* stream) {
    X3D_CHECK_ARG(x && w && y && N > 0 && N <= 65535 && T > 0 && T <= 65535 && H > 0 && W > 0 && Cout > 0);
    if (Cin != 3) { x3d_set_error("stem133: only n_input_channels=3 is built (got %d)", Cin); return X3D_EINVAL; }
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(stem133_fwd_kernel<3>, dim3(cdiv(Ho * Wo, 256), T, N), dim3(256), 0, (hipStream_t)stream, x, w,
                       y, Cout, T, H, W, Ho, Wo);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_stem_wgrad_groups(int N, int T) {
    const int cap = x3d_opt(X3D_OPT_STEM_WG_CAP);
    const int g = N * T * 8;
    return g < cap ? g : cap;
}

extern "C" int x3d_stem133_bwd_weight(const float* x, const float* dy, float* wpartial, int N, int Cin, int Cout,
                                      int T, int H, int W, void* stream) {
    X3D_CHECK_ARG(x && dy && wpartial && N > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(Cin * 9 <= 32 && Cout <= 32);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int groups = x3d_stem_wgrad_groups(N, T);
    const int total = N * T * Ho * cdiv(Wo, SW_PT);
    if (groups > total) groups = total;
    // every group slot of wpartial must be written: launch exactly x3d_stem_wgrad_groups blocks
    groups = x3d_stem_wgrad_groups(N, T);
    hipLaunchKernelGGL(stem133_wgrad_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, x, dy, wpartial, N, Cin,
                       Cout, T, H, W, Ho, Wo, groups);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_dw5t_tiles(int HW) { return cdiv(HW, T5_TILE); }

extern "C" int x3d_dw5t_fwd(const float* x, const float* w, float* y, int N, int C, int T, int HW, float* partial,
                            void* stream) {
    X3D_CHECK_ARG(x && w && y && N > 0 && C > 0 && T > 0 && HW > 0);
    const int tiles = cdiv(HW, T5_TILE);
    X3D_CHECK_ARG((long long)N * C * tiles <= 0x7fffffffLL);
    dim3 grid((unsigned)(N * C * tiles)), block(256);
    if (HW % 4 == 0)
        hipLaunchKernelGGL(dw5t_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, x, w, y, C, T, HW, partial, tiles);
    else
        hipLaunchKernelGGL(dw5t_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, x, w, y, C, T, HW, partial, tiles);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_dw5t_bwd(const float* g, const float* a, const float* cb, const float* w, const float* x,
                            float* dx, float* wpartial, int N, int C, int T, int HW, void* stream) {
    X3D_CHECK_ARG(g && a && cb && w && x && dx && wpartial && N > 0 && C > 0 && T > 0 && HW > 0);
    const int tiles = cdiv(HW, T5_TILE);
    X3D_CHECK_ARG((long long)N * C * tiles <= 0x7fffffffLL);
    dim3 grid((unsigned)(N * C * tiles)), block(256);
    if (HW % 4 == 0)
        hipLaunchKernelGGL(dw5t_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, g, a, cb, w, x, dx, wpartial, C, T,
                           HW, tiles);
    else
        hipLaunchKernelGGL(dw5t_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, g, a, cb, w, x, dx, wpartial, C,
                           T, HW, tiles);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
