// Stem: conv1_s (dense 1x3x3, stride (1,2,2), 3 -> 24; x3d.py:196-201,317) and conv1_t
// (depthwise temporal 5x1x1, pad (2,0,0); x3d.py:202-208,318), forward and backward.
// No BN sits between the two convolutions, so conv1_s emits raw output only; conv1_t's
// epilogue carries the statistics for bn1 (x3d.py:209,319).
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
// j = ci*9 + kh*3 + kw (27 -> 32), co (24 -> 32).  Tiles of 128 voxels are staged in LDS
// ([32][132] each, im2col built on the fly); each of the 4 waves owns one 16x16 tile of dW.
// ---------------------------------------------------------------------------------------
constexpr int SW_PT = 128, SW_LD = 132;

__global__ __launch_bounds__(256) void stem133_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ wpartial, int N, int Cin, int Cout,
                                                            int T, int H, int W, int Ho, int Wo, int groups) {
    __shared__ __attribute__((aligned(16))) float Ld[32 * SW_LD];
    __shared__ __attribute__((aligned(16))) float Lx[32 * SW_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int wr = wave >> 1, wc = wave & 1;
    const int J = Cin * 9;
    const int HWo = Ho * Wo;
    const int tiles_per_plane = cdiv(HWo, SW_PT);
    const int total = N * T * tiles_per_plane;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Staging: thread owns voxel column pp = tid & 127 of the tile and the 16 rows r0, r0 + 2, ... of both LDS images.
    // All 32 loads of a tile (16 dy values, 16 gathered input taps) are issued together from clamped addresses and masked
    // when written to LDS; the loads of the NEXT tile are in flight during the MFMAs of the current one.  (The first
    // version loaded one element per iteration under bounds branches: 16 serial round trips per tile, 224 us per step at
    // the very end of the backward pass where nothing overlaps it.)
    const int pp = tid & (SW_PT - 1), r0 = tid >> 7;
    float dv[16], xv[16];
    bool dok[16], xok[16];
    auto load_tile = [&](int tl) {
        const int plane = tl / tiles_per_plane, pt = (tl - plane * tiles_per_plane) * SW_PT;
        const int n = plane / T, t = plane - n * T;
        const int p = pt + pp;
        const bool pv = p < HWo;
        const int pc = pv ? p : 0;
        const int ho = pc / Wo, wo = pc - ho * Wo;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 2 * i + r0;
            const int rc = min(row, Cout - 1);
            dv[i] = dy[(((size_t)n * Cout + rc) * T + t) * (size_t)HWo + pc];
            dok[i] = pv && row < Cout;
            const int rj = min(row, J - 1);
            const int ci = rj / 9, k = rj - ci * 9, kh = k / 3, kw = k - kh * 3;
            const int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
            const bool inb = hi >= 0 && hi < H && wi >= 0 && wi < W;
            const int hic = min(max(hi, 0), H - 1), wic = min(max(wi, 0), W - 1);
            xv[i] = x[(((size_t)n * Cin + ci) * T + t) * (size_t)H * W + (size_t)hic * W + wic];
            xok[i] = pv && row < J && inb;
        }
    };
    int tl = blockIdx.x;
    if (tl < total) load_tile(tl);
    for (; tl < total; tl += groups) {
        __syncthreads();                       // previous tile's fragments read
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            Ld[(2 * i + r0) * SW_LD + pp] = dok[i] ? dv[i] : 0.f;
            Lx[(2 * i + r0) * SW_LD + pp] = xok[i] ? xv[i] : 0.f;
        }
        __syncthreads();
        if (tl + groups < total) load_tile(tl + groups);
#pragma unroll 2
        for (int kk = 0; kk < SW_PT / 16; ++kk) {
            const float4 av = *reinterpret_cast<const float4*>(&Ld[(wr * 16 + r) * SW_LD + kk * 16 + 4 * q]);
            const float4 bv = *reinterpret_cast<const float4*>(&Lx[(wc * 16 + r) * SW_LD + kk * 16 + 4 * q]);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
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
    const int g = N * T * 4;
    return g < 512 ? g : 512;
}

extern "C" int x3d_stem133_bwd_weight(const float* x, const float* dy, float* wpartial, int N, int Cin, int Cout,
                                      int T, int H, int W, void* stream) {
    X3D_CHECK_ARG(x && dy && wpartial && N > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(Cin * 9 <= 32 && Cout <= 32);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int groups = x3d_stem_wgrad_groups(N, T);
    const int total = N * T * cdiv(Ho * Wo, SW_PT);
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
