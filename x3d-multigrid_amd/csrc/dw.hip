// Channelwise 3x3x3 convolution (conv3x3x3, x3d.py:87-95; Bottleneck.conv2 x3d.py:114,150):
// forward and a fused backward (data + weight).  HBM-bound (4.7 FLOP/B): every input voxel is
// read from HBM once and every output voxel written once.
//
// Work decomposition: one workgroup owns CPB channels x one tile of TH rows (full width) of
// one sample and marches along T.  A ring of four (T-plane) slots in LDS holds the
// activated, zero-padded input rows (BN-apply + ReLU is applied while staging, so padding is
// exact zero in the activated domain, as in nn.Conv3d(padding=1) on the ReLU output).
// Plane t+2 is fetched into registers (float4, W-coalesced) before plane t is computed and
// written to its slot afterwards: HBM latency hides under the stencil arithmetic, one
// barrier per plane.  Each thread produces 4 consecutive outputs along W per plane from
// 9 LDS row reads (b128 + 2 b32); the 27 taps live in registers.
// BN statistics (sum, sum of squares) are accumulated per thread across the whole T march
// and leave as one partial per (sample, channel, tile).
#include "common.h"
#include <map>
#include <mutex>

namespace {

constexpr int DW_TH = 16;      // max rows per tile
constexpr int DW_PADL = 4;     // left pad columns (keeps float4 alignment of data columns)

struct DwGeom {
    int N, C, T, H, W, Ho, Wo, stride;
    int TH, groups, cpb, ipc;     // rows per tile (of the thread-mapped grid), float4 groups per row, channels per block, items per channel
    int IH, WP, slot;             // staged rows per plane, padded row length, floats per slot (cpb*IH*WP)
    int tiles;
};

// thread-mapped grid = OUTPUT grid for forward, INPUT grid for backward
static DwGeom make_geom(int N, int C, int T, int H, int W, int stride, bool backward) {
    DwGeom g;
    g.N = N; g.C = C; g.T = T; g.H = H; g.W = W; g.stride = stride;
    g.Ho = stride == 2 ? (H - 1) / 2 + 1 : H;
    g.Wo = stride == 2 ? (W - 1) / 2 + 1 : W;
    const int GH = backward ? H : g.Ho, GW = backward ? W : g.Wo;   // thread grid
    g.groups = cdiv(GW, 4);
    int th = 256 / g.groups;
    if (th < 1) th = 1;
    if (th > DW_TH) th = DW_TH;
    if (th > GH) th = GH;
    if (backward && stride == 2 && th > 1 && (th & 1)) th -= 1;   // even tile origin in backward stride 2
    g.TH = th;
    g.ipc = th * g.groups;
    int cpb = 256 / g.ipc;
    if (cpb < 1) cpb = 1;
    if (cpb > 16) cpb = 16;
    if (cpb > C) cpb = C;
    g.cpb = cpb;
    g.tiles = cdiv(GH, th);
    const int SW = backward ? g.Wo : W;      // width of the staged tensor
    g.WP = ((SW + 3) / 4) * 4 + 8;
    if (!backward) g.IH = (th - 1) * stride + 3;
    else g.IH = stride == 1 ? th + 2 : th / 2 + 2;
    g.slot = cpb * g.IH * g.WP;
    return g;
}

struct DwFwdArgs {
    const float* x; const float* w; float* y; const float* pre; int pre_act; float* partial;
    DwGeom g;
};

// Fetch one plane (time index t) of the RAW input into registers (loads stay in flight).
// chunk idx -> (cc, ih, w4); NCH chunks per thread.
template <int NCH>
__device__ __forceinline__ void fwd_fetch(const DwFwdArgs& A, int n, int c0, int h_in0, int t, float4 (&reg)[NCH]) {
    const DwGeom& g = A.g;
    const int w4n = g.WP / 4 - 2;                 // data chunks per row (cols 4 .. WP-5)
    const int total = g.cpb * g.IH * w4n;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int idx = i * 256 + threadIdx.x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < total && t >= 0 && t < g.T) {
            const int cc = idx / (g.IH * w4n);
            const int rem = idx - cc * (g.IH * w4n);
            const int ih = rem / w4n, w4 = rem - ih * w4n;
            const int c = c0 + cc, hi = h_in0 + ih, w = w4 * 4;
            if (c < g.C && hi >= 0 && hi < g.H && w < g.W) {
                const float* p = A.x + ((((size_t)n * g.C + c) * g.T + t) * g.H + hi) * (size_t)g.W + w;
                if ((g.W & 3) == 0) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    v.x = p[0];
                    if (w + 1 < g.W) v.y = p[1];
                    if (w + 2 < g.W) v.z = p[2];
                    if (w + 3 < g.W) v.w = p[3];
                }
            }
        }
        reg[i] = v;
    }
}

// Apply the producer's BN + activation and store into the slot; everything outside the tensor
// (padding rows, t out of range, w >= W) is stored as exact zero.
template <int NCH>
__device__ __forceinline__ void fwd_store(const DwFwdArgs& A, int n, int c0, int h_in0, int t, float* slot,
                                          const float4 (&reg)[NCH]) {
    const DwGeom& g = A.g;
    const int w4n = g.WP / 4 - 2;
    const int total = g.cpb * g.IH * w4n;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < total) {
            const int rowi = idx / w4n, w4 = idx - rowi * w4n;     // rowi = cc*IH + ih
            const int cc = rowi / g.IH, ih = rowi - cc * g.IH;
            const int c = c0 + cc, hi = h_in0 + ih, w = w4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < g.T && c < g.C && hi >= 0 && hi < g.H && w < g.W) {
                float sc = 1.f, sh = 0.f;
                if (A.pre != nullptr) { sc = A.pre[((size_t)n * g.C + c) * 2]; sh = A.pre[((size_t)n * g.C + c) * 2 + 1]; }
                v.x = act_fwd(fmaf(sc, reg[i].x, sh), A.pre_act);
                if (w + 1 < g.W) v.y = act_fwd(fmaf(sc, reg[i].y, sh), A.pre_act);
                if (w + 2 < g.W) v.z = act_fwd(fmaf(sc, reg[i].z, sh), A.pre_act);
                if (w + 3 < g.W) v.w = act_fwd(fmaf(sc, reg[i].w, sh), A.pre_act);
            }
            *reinterpret_cast<float4*>(slot + (size_t)rowi * g.WP + DW_PADL + w4 * 4) = v;
        }
    }
}

template <int NCH, int STRIDE>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwFwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const DwGeom& g = A.g;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, c0 = blockIdx.y * g.cpb, n = blockIdx.z;
    const int ho0 = tile * g.TH;
    const int h_in0 = ho0 * STRIDE - 1;
    float* ring = lds;                                  // 4 slots
    float* redbuf = lds + 4 * (size_t)g.slot;           // 256*2 floats

    // zero the whole ring once (halo columns stay zero for the entire march)
    for (int i = tid; i < 4 * g.slot; i += 256) ring[i] = 0.f;

    // item of this thread
    const bool active = tid < g.cpb * g.ipc;
    const int cc = active ? tid / g.ipc : 0;
    const int ri = active ? tid - cc * g.ipc : 0;
    const int row = ri / g.groups, grp = ri - row * g.groups;
    const int c = c0 + cc;
    const bool valid = active && c < g.C && (ho0 + row) < g.Ho;

    float wt[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wt[k] = valid ? A.w[(size_t)c * 27 + k] : 0.f;

    float4 reg[NCH];
    __syncthreads();
    // prologue: planes 0 and 1 (plane -1 is all zero: slot 3 stays zero)
    fwd_fetch<NCH>(A, n, c0, h_in0, 0, reg);
    fwd_store<NCH>(A, n, c0, h_in0, 0, ring + 0 * (size_t)g.slot, reg);
    fwd_fetch<NCH>(A, n, c0, h_in0, 1, reg);
    fwd_store<NCH>(A, n, c0, h_in0, 1, ring + 1 * (size_t)g.slot, reg);
    __syncthreads();

    float s1 = 0.f, s2 = 0.f;
    const int lrow = row * STRIDE;                       // first staged row used by this item
    const int lcol = DW_PADL + grp * 4 * STRIDE - 1;     // first staged column used (w = -1 at PADL-1)
    const size_t ybase = (((size_t)n * g.C + c) * g.T) * (size_t)g.Ho * g.Wo + (size_t)(ho0 + row) * g.Wo + grp * 4;

    for (int t = 0; t < g.T; ++t) {
        fwd_fetch<NCH>(A, n, c0, h_in0, t + 2, reg);     // in flight during the stencil
        if (valid) {
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const float* sl = ring + (size_t)((t + kt + 3) & 3) * g.slot + (size_t)cc * g.IH * g.WP;   // plane t-1+kt
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* rp = sl + (size_t)(lrow + kh) * g.WP + lcol;
                    const float w0 = wt[kt * 9 + kh * 3], w1 = wt[kt * 9 + kh * 3 + 1], w2 = wt[kt * 9 + kh * 3 + 2];
                    if (STRIDE == 1) {
                        const float v0 = rp[0];
                        const float4 m = *reinterpret_cast<const float4*>(rp + 1);
                        const float v5 = rp[5];
                        o0 = fmaf(w0, v0, fmaf(w1, m.x, fmaf(w2, m.y, o0)));
                        o1 = fmaf(w0, m.x, fmaf(w1, m.y, fmaf(w2, m.z, o1)));
                        o2 = fmaf(w0, m.y, fmaf(w1, m.z, fmaf(w2, m.w, o2)));
                        o3 = fmaf(w0, m.z, fmaf(w1, m.w, fmaf(w2, v5, o3)));
                    } else {
                        const float v0 = rp[0];
                        const float4 a = *reinterpret_cast<const float4*>(rp + 1);
                        const float4 b = *reinterpret_cast<const float4*>(rp + 5);
                        o0 = fmaf(w0, v0, fmaf(w1, a.x, fmaf(w2, a.y, o0)));
                        o1 = fmaf(w0, a.y, fmaf(w1, a.z, fmaf(w2, a.w, o1)));
                        o2 = fmaf(w0, a.w, fmaf(w1, b.x, fmaf(w2, b.y, o2)));
                        o3 = fmaf(w0, b.y, fmaf(w1, b.z, fmaf(w2, b.w, o3)));
                    }
                }
            }
            float* py = A.y + ybase + (size_t)t * g.Ho * g.Wo;
            const int wo = grp * 4;
            if ((g.Wo & 3) == 0) {
                *reinterpret_cast<float4*>(py) = make_float4(o0, o1, o2, o3);
            } else {
                py[0] = o0;
                if (wo + 1 < g.Wo) py[1] = o1; else o1 = 0.f;
                if (wo + 2 < g.Wo) py[2] = o2; else o2 = 0.f;
                if (wo + 3 < g.Wo) py[3] = o3; else o3 = 0.f;
            }
            s1 += (o0 + o1) + (o2 + o3);
            s2 = fmaf(o0, o0, fmaf(o1, o1, fmaf(o2, o2, fmaf(o3, o3, s2))));
        }
        fwd_store<NCH>(A, n, c0, h_in0, t + 2, ring + (size_t)((t + 2) & 3) * g.slot, reg);
        __syncthreads();
    }

    if (A.partial != nullptr) {
        redbuf[tid * 2] = valid ? s1 : 0.f;
        redbuf[tid * 2 + 1] = valid ? s2 : 0.f;
        __syncthreads();
        if (tid < g.cpb * 2) {
            const int ch = tid >> 1, which = tid & 1;
            if (c0 + ch < g.C) {
                float s = 0.f;
                for (int i = 0; i < g.ipc; ++i) s += redbuf[(ch * g.ipc + i) * 2 + which];
                A.partial[(((size_t)n * g.C + c0 + ch) * g.tiles + tile) * 2 + which] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Fused backward.  Thread grid = INPUT voxels (4 consecutive w per thread); the ring holds
// dY = cb0*g + cb1*a + cb2 at OUTPUT resolution, zero outside the tensor.
// ---------------------------------------------------------------------------------------
struct DwBwdArgs {
    const float* g; const float* a; const float* cb; const float* w;
    const float* x; const float* pre; int pre_act;
    float* out; float* wpartial; float* partial;
    DwGeom geo;
};

template <int NCH>
__device__ __forceinline__ void bwd_fetch(const DwBwdArgs& A, int n, int c0, int ho_lo, int t, float4 (&rg)[NCH],
                                          float4 (&ra)[NCH]) {
    const DwGeom& g = A.geo;
    const int w4n = g.WP / 4 - 2;
    const int total = g.cpb * g.IH * w4n;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int idx = i * 256 + threadIdx.x;
        float4 vg = make_float4(0.f, 0.f, 0.f, 0.f), va = vg;
        if (idx < total && t >= 0 && t < g.T) {
            const int cc = idx / (g.IH * w4n);
            const int rem = idx - cc * (g.IH * w4n);
            const int ih = rem / w4n, w4 = rem - ih * w4n;
            const int c = c0 + cc, ho = ho_lo + ih, w = w4 * 4;
            if (c < g.C && ho >= 0 && ho < g.Ho && w < g.Wo) {
                const size_t off = ((((size_t)n * g.C + c) * g.T + t) * g.Ho + ho) * (size_t)g.Wo + w;
                if ((g.Wo & 3) == 0) {
                    vg = *reinterpret_cast<const float4*>(A.g + off);
                    va = *reinterpret_cast<const float4*>(A.a + off);
                } else {
                    vg.x = A.g[off]; va.x = A.a[off];
                    if (w + 1 < g.Wo) { vg.y = A.g[off + 1]; va.y = A.a[off + 1]; }
                    if (w + 2 < g.Wo) { vg.z = A.g[off + 2]; va.z = A.a[off + 2]; }
                    if (w + 3 < g.Wo) { vg.w = A.g[off + 3]; va.w = A.a[off + 3]; }
                }
            }
        }
        rg[i] = vg;
        ra[i] = va;
    }
}

// combine to dY and store into the slot (zero outside the tensor, including the w >= Wo tail of a chunk)
template <int NCH>
__device__ __forceinline__ void bwd_store(const DwBwdArgs& A, int n, int c0, int ho_lo, int t, float* slot,
                                          const float4 (&rg)[NCH], const float4 (&ra)[NCH]) {
    const DwGeom& g = A.geo;
    const int w4n = g.WP / 4 - 2;
    const int total = g.cpb * g.IH * w4n;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int idx = i * 256 + threadIdx.x;
        if (idx < total) {
            const int rowi = idx / w4n, w4 = idx - rowi * w4n;
            const int cc = rowi / g.IH, ih = rowi - cc * g.IH;
            const int c = c0 + cc, ho = ho_lo + ih, w = w4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < g.T && c < g.C && ho >= 0 && ho < g.Ho && w < g.Wo) {
                const float* cb = A.cb + ((size_t)n * g.C + c) * 3;
                const float k0 = cb[0], k1 = cb[1], k2 = cb[2];
                v.x = fmaf(k0, rg[i].x, fmaf(k1, ra[i].x, k2));
                if (w + 1 < g.Wo) v.y = fmaf(k0, rg[i].y, fmaf(k1, ra[i].y, k2));
                if (w + 2 < g.Wo) v.z = fmaf(k0, rg[i].z, fmaf(k1, ra[i].z, k2));
                if (w + 3 < g.Wo) v.w = fmaf(k0, rg[i].w, fmaf(k1, ra[i].w, k2));
            }
            *reinterpret_cast<float4*>(slot + (size_t)rowi * g.WP + DW_PADL + w4 * 4) = v;
        }
    }
}

template <int NCH, int STRIDE>
__global__ __launch_bounds__(256) void dw_bwd_kernel(const DwBwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const DwGeom& g = A.geo;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, c0 = blockIdx.y * g.cpb, n = blockIdx.z;
    const int h0 = tile * g.TH;                                   // first input row of the tile
    const int ho_lo = STRIDE == 1 ? h0 - 1 : h0 / 2 - 1;          // first staged output row
    float* ring = lds;

    for (int i = tid; i < 4 * g.slot; i += 256) ring[i] = 0.f;

    const bool active = tid < g.cpb * g.ipc;
    const int cc = active ? tid / g.ipc : 0;
    const int ri = active ? tid - cc * g.ipc : 0;
    const int row = ri / g.groups, grp = ri - row * g.groups;
    const int c = c0 + cc, h = h0 + row, w0 = grp * 4;
    const bool valid = active && c < g.C && h < g.H;

    float wt[27], dwacc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) { wt[k] = valid ? A.w[(size_t)c * 27 + k] : 0.f; dwacc[k] = 0.f; }
    float sc = 1.f, sh = 0.f;
    if (valid && A.pre != nullptr) { sc = A.pre[((size_t)n * g.C + c) * 2]; sh = A.pre[((size_t)n * g.C + c) * 2 + 1]; }

    float4 rg[NCH], ra[NCH];
    __syncthreads();
    bwd_fetch<NCH>(A, n, c0, ho_lo, 0, rg, ra);
    bwd_store<NCH>(A, n, c0, ho_lo, 0, ring + 0 * (size_t)g.slot, rg, ra);
    bwd_fetch<NCH>(A, n, c0, ho_lo, 1, rg, ra);
    bwd_store<NCH>(A, n, c0, ho_lo, 1, ring + 1 * (size_t)g.slot, rg, ra);
    __syncthreads();

    float s1 = 0.f, s2 = 0.f;
    const size_t xbase = (((size_t)n * g.C + c) * g.T) * (size_t)g.H * g.W + (size_t)h * g.W + w0;

    for (int t = 0; t < g.T; ++t) {
        bwd_fetch<NCH>(A, n, c0, ho_lo, t + 2, rg, ra);
        if (valid) {
            // raw forward input of this thread's 4 voxels
            const float* px = A.x + xbase + (size_t)t * g.H * g.W;
            float xv[4] = {0.f, 0.f, 0.f, 0.f};
            if ((g.W & 3) == 0) {
                const float4 q = *reinterpret_cast<const float4*>(px);
                xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (w0 + i < g.W) xv[i] = px[i];
            }
            float hin[4], dact[4], d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s = fmaf(sc, xv[i], sh);
                const bool in = (w0 + i) < g.W;
                hin[i] = in ? act_fwd(s, A.pre_act) : 0.f;
                dact[i] = in ? act_bwd(s, A.pre_act) : 0.f;
            }
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                // plane index tp = t + 1 - kt  -> slot (tp & 3)
                const float* sl = ring + (size_t)((t + 1 - kt + 4) & 3) * g.slot + (size_t)cc * g.IH * g.WP;
                if (STRIDE == 1) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        // output row ho = h + 1 - kh -> staged row (ho - ho_lo) = row + 2 - kh
                        const float* rp = sl + (size_t)(row + 2 - kh) * g.WP + DW_PADL + w0 - 1;
                        float v[6];
                        v[0] = rp[0];
                        const float4 m = *reinterpret_cast<const float4*>(rp + 1);
                        v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
                        v[5] = rp[5];
                        // wo = w + 1 - kw  ->  v index = i + 2 - kw
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wk = wt[kt * 9 + kh * 3 + kw];
                            float acc = dwacc[kt * 9 + kh * 3 + kw];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                d[i] = fmaf(wk, v[i + 2 - kw], d[i]);
                                acc = fmaf(hin[i], v[i + 2 - kw], acc);
                            }
                            dwacc[kt * 9 + kh * 3 + kw] = acc;
                        }
                    }
                } else {
                    // stride 2: (h + 1 - kh) must be even.  h even -> kh = 1; h odd -> kh in {0, 2}
                    // columns: staged cols for wo = 2*grp, 2*grp+1, 2*grp+2
                    const int par = h & 1;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        if (((kh + 1) & 1) == par) {     // (h + 1 - kh) even  <=>  (kh+1) parity == h parity
                            const int ho = (h + 1 - kh) >> 1;
                            const float* rp = sl + (size_t)(ho - ho_lo) * g.WP + DW_PADL + 2 * grp;
                            const float u0 = rp[0], u1 = rp[1], u2 = rp[2];
                            const float k0 = wt[kt * 9 + kh * 3], k1 = wt[kt * 9 + kh * 3 + 1], k2 = wt[kt * 9 + kh * 3 + 2];
                            // i=0 (w even): kw=1, wo=2grp ; i=1: kw=0 -> wo=2grp+1, kw=2 -> wo=2grp
                            // i=2: kw=1, wo=2grp+1      ; i=3: kw=0 -> wo=2grp+2, kw=2 -> wo=2grp+1
                            d[0] = fmaf(k1, u0, d[0]);
                            d[1] = fmaf(k0, u1, fmaf(k2, u0, d[1]));
                            d[2] = fmaf(k1, u1, d[2]);
                            d[3] = fmaf(k0, u2, fmaf(k2, u1, d[3]));
                            dwacc[kt * 9 + kh * 3] = fmaf(hin[1], u1, fmaf(hin[3], u2, dwacc[kt * 9 + kh * 3]));
                            dwacc[kt * 9 + kh * 3 + 1] = fmaf(hin[0], u0, fmaf(hin[2], u1, dwacc[kt * 9 + kh * 3 + 1]));
                            dwacc[kt * 9 + kh * 3 + 2] = fmaf(hin[1], u0, fmaf(hin[3], u1, dwacc[kt * 9 + kh * 3 + 2]));
                        }
                    }
                }
            }
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = d[i] * dact[i];
                s1 += o[i];
                s2 = fmaf(o[i], xv[i], s2);
            }
            float* po = A.out + xbase + (size_t)t * g.H * g.W;
            if ((g.W & 3) == 0) {
                *reinterpret_cast<float4*>(po) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (w0 + i < g.W) po[i] = o[i];
            }
        }
        bwd_store<NCH>(A, n, c0, ho_lo, t + 2, ring + (size_t)((t + 2) & 3) * g.slot, rg, ra);
        __syncthreads();
    }

    // reductions: per channel of the block, over its ipc items, in item order
    float* rb = lds;    // reuse the ring: [29][256]
#pragma unroll
    for (int k = 0; k < 27; ++k) rb[k * 256 + tid] = valid ? dwacc[k] : 0.f;
    rb[27 * 256 + tid] = valid ? s1 : 0.f;
    rb[28 * 256 + tid] = valid ? s2 : 0.f;
    __syncthreads();
    for (int o = tid; o < g.cpb * 29; o += 256) {
        const int ch = o / 29, k = o - ch * 29;
        if (c0 + ch < g.C) {
            float s = 0.f;
            for (int i = 0; i < g.ipc; ++i) s += rb[k * 256 + ch * g.ipc + i];
            const size_t row_id = ((size_t)n * g.C + c0 + ch) * g.tiles + tile;
            if (k < 27) A.wpartial[(((size_t)n * g.tiles + tile) * g.C + c0 + ch) * 27 + k] = s;   // [N][tiles][C][27]
            else if (A.partial != nullptr) A.partial[row_id * 2 + (k - 27)] = s;
        }
    }
}

static size_t fwd_lds_bytes(const DwGeom& g) { return (4 * (size_t)g.slot + 512) * sizeof(float); }
static size_t bwd_lds_bytes(const DwGeom& g) {
    size_t ring = 4 * (size_t)g.slot, red = 29 * 256;
    return (ring > red ? ring : red) * sizeof(float);
}
static int nch_for(const DwGeom& g) { return cdiv(g.cpb * g.IH * (g.WP / 4 - 2), 256); }

}  // namespace

extern "C" int x3d_dw_tiles(int H_out, int W_out) {
    DwGeom g = make_geom(1, 1, 1, H_out, W_out, 1, false);
    return g.tiles;
}

extern "C" int x3d_dw_bwd_tiles(int H, int W, int strideHW) {
    DwGeom g1 = make_geom(1, 1, 1, H, W, strideHW == 2 ? 2 : 1, true);
    return g1.tiles;
}

template <typename K, typename ARGS>
static int dw_launch(K kernel, const ARGS& args, const DwGeom& g, size_t ldsb, hipStream_t s) {
    dim3 grid(g.tiles, cdiv(g.C, g.cpb), g.N), block(256);
    if (ldsb > 48 * 1024) {
        // raise the kernel's dynamic-LDS limit once per (kernel, size) -- not on every launch, so a
        // launch captured into a hipGraph performs no attribute call
        static std::mutex mu;
        static std::map<const void*, size_t> done;
        std::lock_guard<std::mutex> lk(mu);
        const void* key = reinterpret_cast<const void*>(kernel);
        auto it = done.find(key);
        if (it == done.end() || it->second < ldsb) {
            hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            if (e != hipSuccess) { x3d_set_error("hipFuncSetAttribute(%zu): %s", ldsb, hipGetErrorString(e)); return X3D_ELAUNCH; }
            done[key] = ldsb;
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, ldsb, s, args);
    return X3D_OK;
}

#define DW_DISPATCH(KERNEL, ARGS, GEO, LDSB)                                                          \
    do {                                                                                               \
        const int nch = nch_for(GEO);                                                                  \
        int rc_ = X3D_OK;                                                                              \
        if ((GEO).stride == 1) {                                                                       \
            if (nch <= 2) rc_ = dw_launch(KERNEL<2, 1>, ARGS, GEO, LDSB, s);                           \
            else if (nch <= 4) rc_ = dw_launch(KERNEL<4, 1>, ARGS, GEO, LDSB, s);                      \
            else if (nch <= 8) rc_ = dw_launch(KERNEL<8, 1>, ARGS, GEO, LDSB, s);                      \
            else { x3d_set_error("dw333: row too wide (W=%d)", (GEO).W); return X3D_EINVAL; }          \
        } else {                                                                                       \
            if (nch <= 2) rc_ = dw_launch(KERNEL<2, 2>, ARGS, GEO, LDSB, s);                           \
            else if (nch <= 4) rc_ = dw_launch(KERNEL<4, 2>, ARGS, GEO, LDSB, s);                      \
            else if (nch <= 8) rc_ = dw_launch(KERNEL<8, 2>, ARGS, GEO, LDSB, s);                      \
            else { x3d_set_error("dw333: row too wide (W=%d)", (GEO).W); return X3D_EINVAL; }          \
        }                                                                                              \
        if (rc_ != X3D_OK) return rc_;                                                                 \
    } while (0)

extern "C" int x3d_dw333_fwd(const float* x, const float* w, float* y, int N, int C, int T, int H, int W,
                             int strideHW, const float* pre, int pre_act, float* partial, void* stream) {
    X3D_CHECK_ARG(x && w && y);
    X3D_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    DwFwdArgs A;
    A.x = x; A.w = w; A.y = y; A.pre = pre; A.pre_act = pre ? pre_act : X3D_ACT_NONE; A.partial = partial;
    A.g = make_geom(N, C, T, H, W, strideHW, false);
    const size_t ldsb = fwd_lds_bytes(A.g);
    if (ldsb > 160 * 1024) { x3d_set_error("dw333_fwd: LDS tile too large (W=%d)", W); return X3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    DW_DISPATCH(dw_fwd_kernel, A, A.g, ldsb);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_dw333_bwd(const float* g, const float* a, const float* cb, const float* w, const float* x,
                             const float* pre, int pre_act, float* out, float* wpartial, float* partial, int N,
                             int C, int T, int H, int W, int strideHW, void* stream) {
    X3D_CHECK_ARG(g && a && cb && w && x && out && wpartial);
    X3D_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    DwBwdArgs A;
    A.g = g; A.a = a; A.cb = cb; A.w = w; A.x = x; A.pre = pre; A.pre_act = pre ? pre_act : X3D_ACT_NONE;
    A.out = out; A.wpartial = wpartial; A.partial = partial;
    A.geo = make_geom(N, C, T, H, W, strideHW, true);
    const size_t ldsb = bwd_lds_bytes(A.geo);
    if (ldsb > 160 * 1024) { x3d_set_error("dw333_bwd: LDS tile too large (W=%d)", W); return X3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    DW_DISPATCH(dw_bwd_kernel, A, A.geo, ldsb);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
