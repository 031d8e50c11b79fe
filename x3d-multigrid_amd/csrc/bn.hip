// Split-BatchNorm finalisation, squeeze-excitation on pooled statistics, residual epilogue,
// head pooling and fused SGD.
//
// Reference call sites replaced: SubBatchNorm3d.forward (x3d.py:47-58) and its autograd
// backward; SE branch of Bottleneck.forward (x3d.py:153-159); `out += residual; relu`
// (x3d.py:165-169); bn5/relu/avgpool (x3d.py:327-331); torch.optim.SGD step
// (train_x3d_kinetics_multigrid.py:183,277).
//
// None of these touch a full activation tensor except the residual epilogue and the head
// pool: BN statistics arrive as per-(sample, channel, tile) partial sums written by the conv
// epilogues; the kernels here reduce them in fp64 in a fixed order and emit the
// per-(sample, channel) coefficients the next conv applies on load.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------
// tile reduction: partial[N*C][tiles][NV] (float) -> dsum[N*C][NV] (double); one wave per row
// ------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void reduce_tiles_kernel(const float* __restrict__ partial,
                                                           double* __restrict__ dsum, int rows, int tiles) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    double s[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) s[v] = 0.0;
    const float* p = partial + (size_t)row * tiles * NV;
    for (int t = lane; t < tiles; t += 64) {
#pragma unroll
        for (int v = 0; v < NV; ++v) s[v] += (double)p[(size_t)t * NV + v];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double x = s[v];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (lane == 0) dsum[(size_t)row * NV + v] = x;
    }
}

// ------------------------------------------------------------------------------------
// BN forward finalize (thread per channel)
// ------------------------------------------------------------------------------------
__global__ void bn_fwd_finalize_kernel(const double* __restrict__ dsum, int N, int C, int S, int count,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                       float eps, float* __restrict__ coef, float* __restrict__ save,
                                       float* __restrict__ nsum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double cnt = (double)count * (double)(N / S);
    const float g = gamma[c], b = beta[c];
    for (int j = 0; j < S; ++j) {
        double s1 = 0.0, s2 = 0.0;
        for (int n = j; n < N; n += S) {
            s1 += dsum[((size_t)n * C + c) * 2];
            s2 += dsum[((size_t)n * C + c) * 2 + 1];
        }
        const double mean = s1 / cnt;
        double var = s2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        save[(size_t)j * C + c] = (float)mean;
        save[(size_t)(S + j) * C + c] = (float)invstd;
        if (rmean != nullptr) {
            const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            rmean[(size_t)j * C + c] = (float)((1.0 - momentum) * rmean[(size_t)j * C + c] + momentum * mean);
            rvar[(size_t)j * C + c] = (float)((1.0 - momentum) * rvar[(size_t)j * C + c] + momentum * unb);
        }
        const float sc = (float)((double)g * invstd);
        const float sh = (float)((double)b - mean * (double)g * invstd);
        for (int n = j; n < N; n += S) {
            coef[((size_t)n * C + c) * 2] = sc;
            coef[((size_t)n * C + c) * 2 + 1] = sh;
            if (nsum != nullptr) nsum[(size_t)n * C + c] = (float)dsum[((size_t)n * C + c) * 2];
        }
    }
}

__global__ void bn_eval_coef_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                    int N, int C, float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int c = i % C;
    const double invstd = 1.0 / sqrt((double)rvar[c] + (double)eps);
    coef[(size_t)i * 2] = (float)((double)gamma[c] * invstd);
    coef[(size_t)i * 2 + 1] = (float)((double)beta[c] - (double)rmean[c] * (double)gamma[c] * invstd);
}

// ------------------------------------------------------------------------------------
// Fused tile reduction + BN finalize: one workgroup per channel.  For every sample the 256
// threads sum that (n, c) row's tile partials in fp64 (fixed order: thread-strided, then a
// fixed tree), thread 0 adds the row sum to its split; then one thread per split finishes.
// ------------------------------------------------------------------------------------
constexpr int BN_MAXS = 256;
constexpr int BN_MAXN = 1024;

// Per-sample row sums for one channel.  A group of G lanes (G = tiles rounded up to a power of two, <= 64) owns one
// sample, so a wave reduces 64 / G samples per pass (the depthwise partials have 1-4 tiles per sample, the large-batch
// multigrid shapes up to 128 samples); lanes stride the tiles; fp64 butterfly inside the group.  Lanes beyond `tiles`
// contribute exact zeros, so the result does not depend on G.
__device__ __forceinline__ void channel_row_sums(const float* __restrict__ partial, int N, int C, int c, int tiles,
                                                 double* rows /* [N][2] in LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int G = 1;
    while (G < tiles && G < 64) G <<= 1;
    const int spw = 64 / G, sub = lane / G, l = lane - sub * G;
    for (int n0 = wave * spw; n0 < N; n0 += 4 * spw) {
        const int n = n0 + sub;
        const bool ok = n < N;
        const float* p = partial + ((size_t)(ok ? n : 0) * C + c) * tiles * 2;      // clamped: no load under a lane branch
        // four loads in flight per lane (a row of 784 tile pairs was twelve serial round trips); fixed order
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        int t = l;
        for (; t + 3 * G < tiles; t += 4 * G) {
            const float2 v0 = *reinterpret_cast<const float2*>(p + 2 * t);
            const float2 v1 = *reinterpret_cast<const float2*>(p + 2 * (t + G));
            const float2 v2 = *reinterpret_cast<const float2*>(p + 2 * (t + 2 * G));
            const float2 v3 = *reinterpret_cast<const float2*>(p + 2 * (t + 3 * G));
            a0 += (double)v0.x; b0 += (double)v0.y;
            a1 += (double)v1.x; b1 += (double)v1.y;
            a2 += (double)v2.x; b2 += (double)v2.y;
            a3 += (double)v3.x; b3 += (double)v3.y;
        }
        for (; t < tiles; t += G) {
            const float2 v = *reinterpret_cast<const float2*>(p + 2 * t);
            a0 += (double)v.x;
            b0 += (double)v.y;
        }
        double a = (a0 + a1) + (a2 + a3), b = (b0 + b1) + (b2 + b3);
        for (int o = G >> 1; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (l == 0 && ok) { rows[2 * n] = a; rows[2 * n + 1] = b; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void bn_fwd_fused_kernel(const float* __restrict__ partial, int N, int C, int tiles,
                                                           int S, int count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, float momentum, float eps,
                                                           float* __restrict__ coef, float* __restrict__ save,
                                                           float* __restrict__ nsum) {
    __shared__ double rows[2 * BN_MAXN];
    __shared__ float csc[BN_MAXS], csh[BN_MAXS];
    const int c = blockIdx.x, tid = threadIdx.x;
    // (round 4) everything that does not depend on the statistics is requested in front of them: these finalize launches are
    // chains of exposed memory round trips (~1 us each, the producer's data sits in another XCD's L2 / HBM), not arithmetic
    const float gm = gamma[c], bt = beta[c];
    const int j0 = tid < S ? tid : 0;
    const float rm0 = rmean != nullptr ? rmean[(size_t)j0 * C + c] : 0.f, rv0 = rvar != nullptr ? rvar[(size_t)j0 * C + c] : 0.f;
    channel_row_sums(partial, N, C, c, tiles, rows);
    const double cnt = (double)count * (double)(N / S);
    for (int j = tid; j < S; j += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int n = j; n < N; n += S) { s1 += rows[2 * n]; s2 += rows[2 * n + 1]; }
        const double mean = s1 / cnt;
        double var = s2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        save[(size_t)j * C + c] = (float)mean;
        save[(size_t)(S + j) * C + c] = (float)invstd;
        if (rmean != nullptr) {
            const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            const float rmj = j == j0 ? rm0 : rmean[(size_t)j * C + c], rvj = j == j0 ? rv0 : rvar[(size_t)j * C + c];
            rmean[(size_t)j * C + c] = (float)((1.0 - momentum) * rmj + momentum * mean);
            rvar[(size_t)j * C + c] = (float)((1.0 - momentum) * rvj + momentum * unb);
        }
        csc[j] = (float)((double)gm * invstd);
        csh[j] = (float)((double)bt - mean * (double)gm * invstd);
    }
    __syncthreads();
    for (int n = tid; n < N; n += 256) {
        coef[((size_t)n * C + c) * 2] = csc[n % S];
        coef[((size_t)n * C + c) * 2 + 1] = csh[n % S];
        if (nsum != nullptr) nsum[(size_t)n * C + c] = (float)rows[2 * n];
    }
}

__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(const float* __restrict__ partial, int N, int C, int tiles,
                                                           int S, int count, const float* __restrict__ gamma,
                                                           const float* __restrict__ save, float* __restrict__ cb,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int accumulate) {
    __shared__ double rows[2 * BN_MAXN];
    __shared__ float cA[BN_MAXS], cB[BN_MAXS], cC[BN_MAXS];
    __shared__ double dgs[BN_MAXS], dbs[BN_MAXS];
    const int c = blockIdx.x, tid = threadIdx.x;
    const double g = gamma[c];                           // (requested in front of the statistics: see bn_fwd_fused_kernel)
    const int j0 = tid < S ? tid : 0;
    const float mean0 = save[(size_t)j0 * C + c], invstd0 = save[(size_t)(S + j0) * C + c];
    channel_row_sums(partial, N, C, c, tiles, rows);
    const double M = (double)count * (double)(N / S);
    for (int j = tid; j < S; j += 256) {
        double sg = 0.0, sga = 0.0;
        for (int n = j; n < N; n += S) { sg += rows[2 * n]; sga += rows[2 * n + 1]; }
        const double mean = j == j0 ? mean0 : save[(size_t)j * C + c], invstd = j == j0 ? invstd0 : save[(size_t)(S + j) * C + c];
        const double sgx = (sga - mean * sg) * invstd;
        const double k = g * invstd;
        cA[j] = (float)k;
        cB[j] = (float)(-k * invstd * sgx / M);
        cC[j] = (float)(-k * sg / M + k * invstd * mean * sgx / M);
        dgs[j] = sgx;
        dbs[j] = sg;
    }
    __syncthreads();
    for (int n = tid; n < N; n += 256) {
        cb[((size_t)n * C + c) * 3] = cA[n % S];
        cb[((size_t)n * C + c) * 3 + 1] = cB[n % S];
        cb[((size_t)n * C + c) * 3 + 2] = cC[n % S];
    }
    if (tid == 0) {
        double dg = 0.0, db = 0.0;
        for (int j = 0; j < S; ++j) { dg += dgs[j]; db += dbs[j]; }
        if (accumulate) { dgamma[c] += (float)dg; dbeta[c] += (float)db; }
        else { dgamma[c] = (float)dg; dbeta[c] = (float)db; }
    }
}

// ------------------------------------------------------------------------------------
// SE forward: one workgroup per sample
// ------------------------------------------------------------------------------------
constexpr int SE_MAXC = 1024, SE_MAXW = 64;
constexpr int SE_PB = 16;          // hidden units per block-reduction round of the SE kernels

__global__ __launch_bounds__(256) void se_fwd_kernel(const float* __restrict__ coef, const float* __restrict__ nsum,
                                                     int C, int Wd, int count, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, float* __restrict__ coef_out,
                                                     float* __restrict__ save_se, float* __restrict__ save_z,
                                                     float* __restrict__ save_pool) {
    __shared__ float pool[SE_MAXC];
    __shared__ float z[SE_MAXW];
    __shared__ float part[SE_PB][4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        const float sc = coef[((size_t)n * C + c) * 2], sh = coef[((size_t)n * C + c) * 2 + 1];
        const float p = fmaf(sc, nsum[(size_t)n * C + c] / (float)count, sh);
        pool[c] = p;
        save_pool[(size_t)n * C + c] = p;
    }
    __syncthreads();
    // fc1: SE_PB hidden units per round; a thread owns channels tid, tid + 256, ... and issues every weight load of the
    // round at once (one memory round trip per round instead of one per hidden unit), then a fixed-order block sum
    for (int w0 = 0; w0 < Wd; w0 += SE_PB) {
        const float bv = (tid < SE_PB && w0 + tid < Wd) ? b1[w0 + tid] : 0.f;
        float p[SE_PB];
#pragma unroll
        for (int u = 0; u < SE_PB; ++u) p[u] = 0.f;
        for (int c = tid; c < C; c += 256) {
            const float pc = pool[c];
#pragma unroll
            for (int u = 0; u < SE_PB; ++u) p[u] = fmaf(w1[(size_t)min(w0 + u, Wd - 1) * C + c], pc, p[u]);
        }
#pragma unroll
        for (int u = 0; u < SE_PB; ++u) {
            const float t = wave_sum(p[u]);
            if (lane == 0) part[u][wave] = t;
        }
        __syncthreads();
        if (tid < SE_PB && w0 + tid < Wd) {
            float t = (part[tid][0] + part[tid][1]) + (part[tid][2] + part[tid][3]) + bv;
            t = t > 0.f ? t : 0.f;
            z[w0 + tid] = t;
            save_z[(size_t)n * Wd + w0 + tid] = t;
        }
        __syncthreads();
    }
    for (int c = tid; c < C; c += 256) {
        float s = b2[c];
        if ((Wd & 3) == 0) {          // 16-byte loads along the channel's row (see se_bwd_sample_kernel)
            for (int w = 0; w < Wd; w += 4) {
                const float4 v = *reinterpret_cast<const float4*>(w2 + (size_t)c * Wd + w);
                s = fmaf(v.x, z[w], s); s = fmaf(v.y, z[w + 1], s); s = fmaf(v.z, z[w + 2], s); s = fmaf(v.w, z[w + 3], s);
            }
        } else {
            for (int w = 0; w < Wd; ++w) s = fmaf(w2[(size_t)c * Wd + w], z[w], s);
        }
        const float se = sigmoidf_(s);
        save_se[(size_t)n * C + c] = se;
        coef_out[((size_t)n * C + c) * 2] = coef[((size_t)n * C + c) * 2] * se;
        coef_out[((size_t)n * C + c) * 2 + 1] = coef[((size_t)n * C + c) * 2 + 1] * se;
    }
}

// ------------------------------------------------------------------------------------
// Round 4: BN2's training finalize AND the SE branch in ONE launch (x3d.py:151-159 behind SubBatchNorm3d.forward :47-58).
// bn_fwd_fused_kernel (workgroup per channel) + se_fwd_kernel (workgroup per sample, 256 threads, five serial memory round
// trips: coefficients, two fc1 rounds, fc2) were two dependent launches of 4.8 + 6-12 us for microseconds of arithmetic;
// every dependent launch of the replayed graph costs ~4.7 us whatever it does (profiles/r04/a_chain.txt).
// Workgroup = sample n, 1024 threads, thread = channel: each workgroup reduces the statistics of ITS BN split redundantly
// (the channelwise conv leaves 1-4 tile pairs per (sample, channel): N/S x tiles x 8 B per thread) in fp64, fixed order --
// every workgroup of a split computes bitwise the same mean / invstd; the split's first sample publishes them and updates
// the running statistics.  All weights (fc1 rows for this wave's hidden units, fc2 row of this thread's channel) are
// requested in front of the statistics loads: ONE exposed memory round trip, then LDS / DPP arithmetic only.
// UPW: hidden units per wave (Wd <= 16 UPW); CPL: channels per lane of an fc1 row (C <= 64 CPL).
// ------------------------------------------------------------------------------------
struct SeBnArgs {
    const float* partial; const float* gamma; const float* beta; float* rmean; float* rvar;
    const float* w1; const float* b1; const float* w2; const float* b2;
    float* coef_out; float* save; float* nsum; float* save_se; float* save_z; float* save_pool;
    int N, C, tiles, S, count, Wd;
    float momentum, eps;
};

template <int UPW, int CPL>
__global__ __launch_bounds__(1024) void se_bn_fwd_kernel(const SeBnArgs A) {
    __shared__ float pool[SE_MAXC];
    __shared__ float z[SE_MAXW];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = A.C, Wd = A.Wd, S = A.S, tiles = A.tiles;
    const int j = n % S, ns = A.N / S;
    const bool cv = tid < C;
    const int c = cv ? tid : C - 1;                          // clamped: no load under a lane branch
    // ---- (1) weights: independent of the statistics, requested first
    float w1r[UPW][CPL];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = min(wave + 16 * i, Wd - 1);
#pragma unroll
        for (int k = 0; k < CPL; ++k) w1r[i][k] = A.w1[(size_t)u * C + min(lane + 64 * k, C - 1)];
    }
    float w2r[16 * UPW];
    if ((Wd & 3) == 0) {
#pragma unroll
        for (int q4 = 0; q4 < 4 * UPW; ++q4) {
            const float4 v = *reinterpret_cast<const float4*>(A.w2 + (size_t)c * Wd + min(4 * q4, Wd - 4));
            w2r[4 * q4] = v.x; w2r[4 * q4 + 1] = v.y; w2r[4 * q4 + 2] = v.z; w2r[4 * q4 + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int w = 0; w < 16 * UPW; ++w) w2r[w] = A.w2[(size_t)c * Wd + min(w, Wd - 1)];
    }
    const float b2c = A.b2[c], gm = A.gamma[c], bt = A.beta[c];
    // ---- (2) statistics of this sample's split: row sums per sample (over its tiles), then over the samples, fp64
    // (fp64 sums of fp32 tile sums: exact as long as the terms' exponents span < 2^29, so the grouping -- tiles outer,
    // samples inner here; per-sample rows first in bn_fwd_fused_kernel -- does not show in the result)
    double s1 = 0.0, s2 = 0.0, own = 0.0;
    for (int m0 = 0; m0 < ns; m0 += 8) {
        for (int t = 0; t < tiles; ++t) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int nn = j + min(m0 + u, ns - 1) * S;
                v[u] = *reinterpret_cast<const float2*>(A.partial + (((size_t)nn * C + c) * tiles + t) * 2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (m0 + u < ns) {
                    s1 += (double)v[u].x; s2 += (double)v[u].y;
                    if (j + (m0 + u) * S == n) own += (double)v[u].x;
                }
            }
        }
    }
    // ---- (3) BN2 coefficients (same arithmetic as bn_fwd_fused_kernel), pooled BN output of this sample
    const double cnt = (double)A.count * (double)ns;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)A.eps);
    const float sc = (float)((double)gm * invstd);
    const float sh = (float)((double)bt - mean * (double)gm * invstd);
    float p = 0.f;
    if (cv) {
        if (n == j) {                                    // first sample of the split publishes its statistics
            A.save[(size_t)j * C + c] = (float)mean;
            A.save[(size_t)(S + j) * C + c] = (float)invstd;
            if (A.rmean != nullptr) {
                const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                A.rmean[(size_t)j * C + c] = (float)((1.0 - A.momentum) * A.rmean[(size_t)j * C + c] + A.momentum * mean);
                A.rvar[(size_t)j * C + c] = (float)((1.0 - A.momentum) * A.rvar[(size_t)j * C + c] + A.momentum * unb);
            }
        }
        const float nsf = (float)own;
        A.nsum[(size_t)n * C + c] = nsf;
        p = fmaf(sc, nsf / (float)A.count, sh);
        A.save_pool[(size_t)n * C + c] = p;
    }
    pool[tid] = p;                                       // (zeros beyond C: the fc1 rows are summed over 64 CPL slots)
    __syncthreads();
    // ---- (4) fc1 + ReLU: wave w owns hidden units w, w + 16, ...
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = wave + 16 * i;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int cc = lane + 64 * k;
            acc = fmaf(cc < C ? w1r[i][k] : 0.f, pool[cc & (SE_MAXC - 1)], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0 && u < Wd) {
            float t = acc + A.b1[u];
            t = t > 0.f ? t : 0.f;
            z[u] = t;
            A.save_z[(size_t)n * Wd + u] = t;
        }
    }
    __syncthreads();
    // ---- (5) fc2 + sigmoid, gate folded into the coefficients conv3 applies on load
    float s = b2c;
#pragma unroll
    for (int w = 0; w < 16 * UPW; ++w) {
        if (w < Wd) s = fmaf(w2r[w], z[w], s);
    }
    if (cv) {
        const float se = sigmoidf_(s);
        A.save_se[(size_t)n * C + c] = se;
        A.coef_out[((size_t)n * C + c) * 2] = sc * se;
        A.coef_out[((size_t)n * C + c) * 2 + 1] = sh * se;
    }
}

// ------------------------------------------------------------------------------------
// BN backward finalize (thread per channel).  dsum[n][c] = {sum g, sum g*raw}.
// extra (optional, SE variant): per-(n,c) arrays modifying the upstream gradient
//   g_full = se[n,c]*g + dpool[n,c]/count
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void bn_bwd_finalize_channel(int c, const double* __restrict__ dsum, int N, int C, int S, int count,
                                       const float* __restrict__ gamma, const float* __restrict__ save,
                                       const float* __restrict__ se, const float* __restrict__ dpool,
                                       const float* __restrict__ nsum, float* __restrict__ cb,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    const double M = (double)count * (double)(N / S);
    const double g = gamma[c];
    double dg = 0.0, db = 0.0;
    for (int j = 0; j < S; ++j) {
        const double mean = save[(size_t)j * C + c], invstd = save[(size_t)(S + j) * C + c];
        double sg = 0.0, sgx = 0.0;
#pragma unroll 4
        for (int n = j; n < N; n += S) {
            const double d0 = dsum[((size_t)n * C + c) * 2], d1 = dsum[((size_t)n * C + c) * 2 + 1];
            if (se != nullptr) {
                const double sv = se[(size_t)n * C + c], dp = dpool[(size_t)n * C + c];
                sg += sv * d0 + dp;
                sgx += sv * (d1 - mean * d0) * invstd +
                       (dp / (double)count) * ((double)nsum[(size_t)n * C + c] - (double)count * mean) * invstd;
            } else {
                sg += d0;
                sgx += (d1 - mean * d0) * invstd;
            }
        }
        dg += sgx;
        db += sg;
        const double k = g * invstd;
        const double Bc = -k * invstd * sgx / M;
        const double Cbase = -k * sg / M + k * invstd * mean * sgx / M;
        for (int n = j; n < N; n += S) {
            double Ac = k, Cc = Cbase;
            if (se != nullptr) {
                Ac = k * (double)se[(size_t)n * C + c];
                Cc += k * (double)dpool[(size_t)n * C + c] / (double)count;
            }
            cb[((size_t)n * C + c) * 3] = (float)Ac;
            cb[((size_t)n * C + c) * 3 + 1] = (float)Bc;
            cb[((size_t)n * C + c) * 3 + 2] = (float)Cc;
        }
    }
    if (accumulate) {
        dgamma[c] += (float)dg;
        dbeta[c] += (float)db;
    } else {
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ dsum, int N, int C, int S, int count,
                                       const float* __restrict__ gamma, const float* __restrict__ save,
                                       const float* __restrict__ se, const float* __restrict__ dpool,
                                       const float* __restrict__ nsum, float* __restrict__ cb,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    bn_bwd_finalize_channel(c, dsum, N, C, S, count, gamma, save, se, dpool, nsum, cb, dgamma, dbeta, accumulate);
}

// SE backward, one workgroup per sample.  Writes dpool[n][c], dz2[n][c], dz1[n][w].
__global__ __launch_bounds__(256) void se_bwd_sample_kernel(
    const double* __restrict__ dsum, int C, int S, int Wd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ save, const float* __restrict__ w1,
    const float* __restrict__ w2, const float* __restrict__ save_se, const float* __restrict__ save_z,
    float* __restrict__ dpool, float* __restrict__ dz2o, float* __restrict__ dz1o) {
    __shared__ float dz2[SE_MAXC];
    __shared__ float dz1[SE_MAXW];
    __shared__ float part[SE_PB][4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = n % S;
    for (int c = tid; c < C; c += 256) {
        const double mean = save[(size_t)j * C + c], invstd = save[(size_t)(S + j) * C + c];
        const double k = (double)gamma[c] * invstd, h = (double)beta[c] - mean * k;
        const double d0 = dsum[((size_t)n * C + c) * 2], d1 = dsum[((size_t)n * C + c) * 2 + 1];
        const float dse = (float)(k * d1 + h * d0);
        const float sv = save_se[(size_t)n * C + c];
        const float v = dse * sv * (1.f - sv);
        dz2[c] = v;
        dz2o[(size_t)n * C + c] = v;
    }
    __syncthreads();
    // fc2 backward onto the hidden units: same round structure as se_fwd_kernel's fc1 (a thread reads SE_PB consecutive
    // floats of its channels' rows of w2)
    for (int w0 = 0; w0 < Wd; w0 += SE_PB) {
        const float zv = (tid < SE_PB && w0 + tid < Wd) ? save_z[(size_t)n * Wd + w0 + tid] : 0.f;
        float p[SE_PB];
#pragma unroll
        for (int u = 0; u < SE_PB; ++u) p[u] = 0.f;
        if ((Wd & 3) == 0) {
            // a thread's 16 weights are consecutive in its channel's row: four 16-byte loads instead of sixteen dword
            // loads that each touch 64 different cache lines (the rows are Wd floats apart across the lanes)
            for (int c = tid; c < C; c += 256) {
                const float dc = dz2[c];
#pragma unroll
                for (int q4 = 0; q4 < SE_PB / 4; ++q4) {
                    const int wq = min(w0 + 4 * q4, Wd - 4);                 // clamped: products of repeated columns land
                    const float4 v = *reinterpret_cast<const float4*>(w2 + (size_t)c * Wd + wq);   // in p[u] with w0+u >= Wd,
                    p[4 * q4] = fmaf(v.x, dc, p[4 * q4]);                    // which are never read
                    p[4 * q4 + 1] = fmaf(v.y, dc, p[4 * q4 + 1]);
                    p[4 * q4 + 2] = fmaf(v.z, dc, p[4 * q4 + 2]);
                    p[4 * q4 + 3] = fmaf(v.w, dc, p[4 * q4 + 3]);
                }
            }
        } else {
            for (int c = tid; c < C; c += 256) {
                const float dc = dz2[c];
#pragma unroll
                for (int u = 0; u < SE_PB; ++u) p[u] = fmaf(w2[(size_t)c * Wd + min(w0 + u, Wd - 1)], dc, p[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < SE_PB; ++u) {
            const float t = wave_sum(p[u]);
            if (lane == 0) part[u][wave] = t;
        }
        __syncthreads();
        if (tid < SE_PB && w0 + tid < Wd) {
            const float t = (part[tid][0] + part[tid][1]) + (part[tid][2] + part[tid][3]);
            const float v = zv > 0.f ? t : 0.f;
            dz1[w0 + tid] = v;
            dz1o[(size_t)n * Wd + w0 + tid] = v;
        }
        __syncthreads();
    }
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int w = 0; w < Wd; ++w) s = fmaf(w1[(size_t)w * C + c], dz1[w], s);
        dpool[(size_t)n * C + c] = s;
    }
}

// Round 4: reduce_tiles_kernel<2> + se_bwd_sample_kernel in ONE launch for the stage 3-4 blocks (98 / 25 statistics tiles per
// row): workgroup = sample, 1024 threads.  (1) every weight the sample needs is requested first (fc2 row and fc1 column of
// the thread's channel: 2 Wd registers); (2) TPR threads per channel row sum its tile pairs in fp64 (lane-strided, four
// loads in flight, butterfly over the TPR lanes: fixed order) -> dsum[n][c] for se_tail_kernel and, through LDS, for
// (3) dz2 = dL/d(fc2 pre-activation); (4) fc2 backward: products w2[c][u] * dz2[c] through an LDS image [16 units][C],
// wave w sums unit w of each 16-unit round lane-strided (fixed order); (5) fc1 backward per channel from registers.
// The old pair cost 4.8 + 6.5-16 us (se_bwd_sample_kernel: 256 threads, one memory round trip per 16 hidden units).
struct SeBwdArgs {
    const float* partial; double* dsum; const float* gamma; const float* beta; const float* save;
    const float* w1; const float* w2; const float* save_se; const float* save_z;
    float* dpool; float* dz2o; float* dz1o;
    int C, S, Wd, tiles, tpr;
};

template <int UPW>
__global__ __launch_bounds__(1024) void se_bwd_sample2_kernel(const SeBwdArgs A) {
    __shared__ double dsl[2 * SE_MAXC];
    __shared__ float dz1s[SE_MAXW];
    __shared__ float prod[16][SE_MAXC];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = A.C, Wd = A.Wd, S = A.S, tiles = A.tiles, TPR = A.tpr;
    const int j = n % S;
    const bool cv = tid < C;
    const int c = cv ? tid : C - 1;
    // ---- (1) weights and per-channel constants of this thread's channel
    float w2r[16 * UPW], w1r[16 * UPW];
    if ((Wd & 3) == 0) {
#pragma unroll
        for (int q4 = 0; q4 < 4 * UPW; ++q4) {
            const float4 v = *reinterpret_cast<const float4*>(A.w2 + (size_t)c * Wd + min(4 * q4, Wd - 4));
            w2r[4 * q4] = v.x; w2r[4 * q4 + 1] = v.y; w2r[4 * q4 + 2] = v.z; w2r[4 * q4 + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int w = 0; w < 16 * UPW; ++w) w2r[w] = A.w2[(size_t)c * Wd + min(w, Wd - 1)];
    }
#pragma unroll
    for (int w = 0; w < 16 * UPW; ++w) w1r[w] = A.w1[(size_t)min(w, Wd - 1) * C + c];
    const double mean = A.save[(size_t)j * C + c], invstd = A.save[(size_t)(S + j) * C + c];
    const double k = (double)A.gamma[c] * invstd, h = (double)A.beta[c] - mean * k;
    const float sv = A.save_se[(size_t)n * C + c];
    // ---- (2) tile sums of this sample's rows: TPR lanes per row (C * TPR <= 1024)
    {
        const int row = tid / TPR, l = tid - row * TPR;
        const bool rv = row < C;
        const float* p = A.partial + ((size_t)n * C + (rv ? row : 0)) * tiles * 2;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        int t = l;
        for (; t + 3 * TPR < tiles; t += 4 * TPR) {
            const float2 v0 = *reinterpret_cast<const float2*>(p + 2 * t);
            const float2 v1 = *reinterpret_cast<const float2*>(p + 2 * (t + TPR));
            const float2 v2 = *reinterpret_cast<const float2*>(p + 2 * (t + 2 * TPR));
            const float2 v3 = *reinterpret_cast<const float2*>(p + 2 * (t + 3 * TPR));
            a0 += (double)v0.x; b0 += (double)v0.y;
            a1 += (double)v1.x; b1 += (double)v1.y;
            a2 += (double)v2.x; b2 += (double)v2.y;
            a3 += (double)v3.x; b3 += (double)v3.y;
        }
        for (; t < tiles; t += TPR) {
            const float2 v = *reinterpret_cast<const float2*>(p + 2 * t);
            a0 += (double)v.x;
            b0 += (double)v.y;
        }
        double a = (a0 + a1) + (a2 + a3), b = (b0 + b1) + (b2 + b3);
        for (int o = TPR >> 1; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (l == 0 && rv) {
            dsl[2 * row] = a; dsl[2 * row + 1] = b;
            A.dsum[((size_t)n * C + row) * 2] = a;
            A.dsum[((size_t)n * C + row) * 2 + 1] = b;
        }
    }
    __syncthreads();
    // ---- (3) dL/d(fc2 pre-activation) of this channel (se_bwd_sample_kernel's arithmetic)
    float dz2v = 0.f;
    if (cv) {
        const double d0 = dsl[2 * c], d1 = dsl[2 * c + 1];
        const float dse = (float)(k * d1 + h * d0);
        dz2v = dse * sv * (1.f - sv);
        A.dz2o[(size_t)n * C + c] = dz2v;
    }
    // ---- (4) fc2 backward onto the hidden units, 16 units per round
    for (int w0 = 0; w0 < Wd; w0 += 16) {
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            if (16 * i == w0) {
#pragma unroll
                for (int u = 0; u < 16; ++u) prod[u][tid] = cv ? w2r[16 * i + u] * dz2v : 0.f;
            }
        }
        __syncthreads();
        const int u = w0 + wave;                         // wave w: unit w0 + w
        float acc = 0.f;
        for (int cc = lane; cc < C; cc += 64) acc += prod[wave][cc];
        acc = wave_sum(acc);
        if (lane == 0 && u < Wd) {
            const float zv = A.save_z[(size_t)n * Wd + u];
            const float v = zv > 0.f ? acc : 0.f;
            dz1s[u] = v;
            A.dz1o[(size_t)n * Wd + u] = v;
        }
        __syncthreads();
    }
    // ---- (5) fc1 backward onto the pooled channel (same order as se_bwd_sample_kernel: w ascending)
    if (cv) {
        float sp = 0.f;
#pragma unroll
        for (int w = 0; w < 16 * UPW; ++w) {
            if (w < Wd) sp = fmaf(w1r[w], dz1s[w], sp);
        }
        A.dpool[(size_t)n * C + c] = sp;
    }
}

// SE weight gradients: thread per (c, w) pair; sums over samples in order.
__device__ __forceinline__ void se_wgrad_element(int i, int N, int C, int Wd, const float* __restrict__ dz2, const float* __restrict__ dz1,
                                const float* __restrict__ save_z, const float* __restrict__ save_pool,
                                float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                float* __restrict__ db2) {
    const int c = i / Wd, w = i - c * Wd;
    double a2 = 0.0, a1 = 0.0, bb2 = 0.0, bb1 = 0.0;
    // samples in batches of eight: the 32 independent loads of a batch are issued together (clamped sample index), then
    // accumulated in sample order -- at N = 128 (multigrid long cycle 0) a one-sample-at-a-time loop is 128 serial round trips
    for (int n0 = 0; n0 < N; n0 += 8) {
        float v2[8], v1[8], vz[8], vp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = min(n0 + u, N - 1);
            v2[u] = dz2[(size_t)n * C + c];
            v1[u] = dz1[(size_t)n * Wd + w];
            vz[u] = save_z[(size_t)n * Wd + w];
            vp[u] = save_pool[(size_t)n * C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (n0 + u < N) {
                const double d2 = v2[u], d1 = v1[u];
                a2 += d2 * (double)vz[u];
                a1 += d1 * (double)vp[u];
                bb2 += d2;
                bb1 += d1;
            }
        }
    }
    dw2[(size_t)c * Wd + w] = (float)a2;
    dw1[(size_t)w * C + c] = (float)a1;
    if (w == 0) db2[c] = (float)bb2;
    if (c == 0) db1[w] = (float)bb1;
}

// Wave-per-channel form of bn_bwd_finalize_channel for the SE-scaled BN (64 % S == 0): lane l owns samples l, l + 64, ...
// (all of split l % S); fp64 butterflies over the lanes of one residue give the split sums, over the residues the
// channel sums.  At N = 128 the thread-per-channel loop is 128 serial round trips of five loads in one or two workgroups.
__device__ __forceinline__ void bn_bwd_finalize_wave(int c, int lane, const double* __restrict__ dsum, int N, int C, int S,
                                                     int count, const float* __restrict__ gamma, const float* __restrict__ save,
                                                     const float* __restrict__ se, const float* __restrict__ dpool,
                                                     const float* __restrict__ nsum, float* __restrict__ cb,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int j = lane % S;
    const double M = (double)count * (double)(N / S);
    const double g = gamma[c];
    const double mean = save[(size_t)j * C + c], invstd = save[(size_t)(S + j) * C + c];
    double sg = 0.0, sgx = 0.0;
    for (int n = lane; n < N; n += 64) {
        const double d0 = dsum[((size_t)n * C + c) * 2], d1 = dsum[((size_t)n * C + c) * 2 + 1];
        const double sv = se[(size_t)n * C + c], dp = dpool[(size_t)n * C + c];
        sg += sv * d0 + dp;
        sgx += sv * (d1 - mean * d0) * invstd +
               (dp / (double)count) * ((double)nsum[(size_t)n * C + c] - (double)count * mean) * invstd;
    }
    for (int o = 32; o >= S; o >>= 1) { sg += __shfl_xor(sg, o); sgx += __shfl_xor(sgx, o); }     // split totals
    double dg = sgx, db = sg;
    for (int o = S >> 1; o > 0; o >>= 1) { dg += __shfl_xor(dg, o); db += __shfl_xor(db, o); }       // channel totals
    const double k = g * invstd;
    const double Bc = -k * invstd * sgx / M;
    const double Cbase = -k * sg / M + k * invstd * mean * sgx / M;
    for (int n = lane; n < N; n += 64) {
        cb[((size_t)n * C + c) * 3] = (float)(k * (double)se[(size_t)n * C + c]);
        cb[((size_t)n * C + c) * 3 + 1] = (float)Bc;
        cb[((size_t)n * C + c) * 3 + 2] = (float)(Cbase + k * (double)dpool[(size_t)n * C + c] / (double)count);
    }
    if (lane == 0) { dgamma[c] = (float)dg; dbeta[c] = (float)db; }
}

// One launch for the two independent consumers of the per-sample SE backward: blocks [0, nbw) compute the SE weight
// gradients, the remaining blocks the BN-backward coefficients / dgamma / dbeta of the SE-scaled BN.
__global__ __launch_bounds__(256) void se_tail_kernel(int nbw, const double* __restrict__ dsum, int N, int C, int S, int count,
                                                      int Wd, const float* __restrict__ gamma, const float* __restrict__ save,
                                                      const float* __restrict__ se, const float* __restrict__ dpool,
                                                      const float* __restrict__ nsum, float* __restrict__ cb,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      const float* __restrict__ dz2, const float* __restrict__ dz1,
                                                      const float* __restrict__ save_z, const float* __restrict__ save_pool,
                                                      float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                      float* __restrict__ db2) {
    if ((int)blockIdx.x < nbw) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < C * Wd) se_wgrad_element(i, N, C, Wd, dz2, dz1, save_z, save_pool, dw1, db1, dw2, db2);
    } else if (64 % S == 0) {
        const int c = ((int)blockIdx.x - nbw) * 4 + (threadIdx.x >> 6);                  // one wave per channel
        if (c < C) bn_bwd_finalize_wave(c, threadIdx.x & 63, dsum, N, C, S, count, gamma, save, se, dpool, nsum, cb, dgamma, dbeta);
    } else {
        const int c = ((int)blockIdx.x - nbw) * 256 + threadIdx.x;
        if (c < C) bn_bwd_finalize_channel(c, dsum, N, C, S, count, gamma, save, se, dpool, nsum, cb, dgamma, dbeta, 0);
    }
}

// ------------------------------------------------------------------------------------
// Residual epilogue, head pooling (elementwise, HBM-bound; float4 when P % 4 == 0)
// ------------------------------------------------------------------------------------
constexpr int EW_TILE = 2048;   // elements per workgroup tile (256 threads x 2 float4)

template <bool VEC>
__global__ __launch_bounds__(256) void bn_add_relu_fwd_kernel(const float* __restrict__ a3, const float* __restrict__ c3,
                                                              const float* __restrict__ res, const float* __restrict__ cd,
                                                              float* __restrict__ out, int P, int ewt) {
    // 1-D grid (rows x tiles, tile fastest): N * C is not bounded by the 65535 limit of grid.y
    const int row = blockIdx.x / ewt, tix = blockIdx.x - row * ewt;
    const float sc = c3[(size_t)row * 2], sh = c3[(size_t)row * 2 + 1];
    float rc = 1.f, rh = 0.f;
    if (cd != nullptr) { rc = cd[(size_t)row * 2]; rh = cd[(size_t)row * 2 + 1]; }
    const size_t base = (size_t)row * P;
    const int p0 = tix * EW_TILE;
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + (u * 256 + threadIdx.x) * 4;
            if (p < P) {
                const float4 a = *reinterpret_cast<const float4*>(a3 + base + p);
                const float4 r = *reinterpret_cast<const float4*>(res + base + p);
                float4 o;
                o.x = fmaxf(fmaf(sc, a.x, sh) + fmaf(rc, r.x, rh), 0.f);
                o.y = fmaxf(fmaf(sc, a.y, sh) + fmaf(rc, r.y, rh), 0.f);
                o.z = fmaxf(fmaf(sc, a.z, sh) + fmaf(rc, r.z, rh), 0.f);
                o.w = fmaxf(fmaf(sc, a.w, sh) + fmaf(rc, r.w, rh), 0.f);
                *reinterpret_cast<float4*>(out + base + p) = o;
            }
        }
    } else {
        for (int p = p0 + threadIdx.x; p < min(P, p0 + EW_TILE); p += 256)
            out[base + p] = fmaxf(fmaf(sc, a3[base + p], sh) + fmaf(rc, res[base + p], rh), 0.f);
    }
}

// Training form of the block output with the BN3 finalize folded in (one launch less per block on the forward chain,
// where kernel times simply add up): every workgroup first requests its slice of a3 / residual, then reduces the
// statistics of ITS (split, channel) from conv3's per-tile partials (N/S samples x tiles pairs, fp64, fixed order -- every
// workgroup of a (split, channel) gets the identical value), derives scale / shift and applies them.  The workgroup with
// blockIdx.x == 0 of the first sample of a split writes mean / invstd for the backward pass and the running statistics
// (x3d.py:47-58, momentum 0.1, unbiased variance).
template <bool VEC>
__global__ __launch_bounds__(256) void bn_stats_add_relu_fwd_kernel(
    const float* __restrict__ a3, const float* __restrict__ partial, int tiles, int N, int C, int S, int count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
    float momentum, float eps, float* __restrict__ save, const float* __restrict__ res, const float* __restrict__ cd,
    float* __restrict__ out, int P, int ewt) {
    __shared__ double redd[4 * 2];
    const int row = blockIdx.x / ewt, tix = blockIdx.x - row * ewt, tid = threadIdx.x;
    const int n = row / C, c = row - n * C, j = n % S;
    const size_t base = (size_t)row * P;
    const int p0 = tix * EW_TILE;
    // 1. the streaming reads do not depend on the statistics: issue them first
    float4 av[2], rv[2];
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + (u * 256 + tid) * 4;
            const int pc = p < P ? p : 0;
            av[u] = *reinterpret_cast<const float4*>(a3 + base + pc);
            rv[u] = *reinterpret_cast<const float4*>(res + base + pc);
        }
    }
    // 2. statistics of (split j, channel c)
    const int ns = N / S, ne = ns * tiles;
    double s1 = 0.0, s2 = 0.0;
    {
        // four pairs in flight per thread (a row of 784-1568 pairs was 4-7 serial round trips in front of the barrier);
        // fixed order: four strided partial sums, then a fixed tree
        auto ldp = [&](int e) -> float2 {
            const int ec = e < ne ? e : 0;
            const int k = ec / tiles, t = ec - k * tiles;
            const float2 v = *reinterpret_cast<const float2*>(partial + (((size_t)(j + k * S) * C + c) * tiles + t) * 2);
            return e < ne ? v : make_float2(0.f, 0.f);
        };
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3_ = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        for (int e = tid; e < ne; e += 1024) {
            const float2 v0 = ldp(e), v1 = ldp(e + 256), v2 = ldp(e + 512), v3 = ldp(e + 768);
            a0 += (double)v0.x; b0 += (double)v0.y;
            a1 += (double)v1.x; b1 += (double)v1.y;
            a2 += (double)v2.x; b2 += (double)v2.y;
            a3_ += (double)v3.x; b3 += (double)v3.y;
        }
        s1 = (a0 + a1) + (a2 + a3_);
        s2 = (b0 + b1) + (b2 + b3);
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((tid & 63) == 0) { redd[(tid >> 6) * 2] = s1; redd[(tid >> 6) * 2 + 1] = s2; }
    __syncthreads();
    s1 = (redd[0] + redd[2]) + (redd[4] + redd[6]);
    s2 = (redd[1] + redd[3]) + (redd[5] + redd[7]);
    const double cnt = (double)count * (double)ns;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * invstd);
    const float sh = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
    if (tix == 0 && n == j && tid == 0) {
        save[(size_t)j * C + c] = (float)mean;
        save[(size_t)(S + j) * C + c] = (float)invstd;
        if (rmean != nullptr) {
            const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            rmean[(size_t)j * C + c] = (float)((1.0 - momentum) * rmean[(size_t)j * C + c] + momentum * mean);
            rvar[(size_t)j * C + c] = (float)((1.0 - momentum) * rvar[(size_t)j * C + c] + momentum * unb);
        }
    }
    float rc = 1.f, rh = 0.f;
    if (cd != nullptr) { rc = cd[(size_t)row * 2]; rh = cd[(size_t)row * 2 + 1]; }
    // 3. elementwise
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + (u * 256 + tid) * 4;
            if (p < P) {
                float4 o;
                o.x = fmaxf(fmaf(sc, av[u].x, sh) + fmaf(rc, rv[u].x, rh), 0.f);
                o.y = fmaxf(fmaf(sc, av[u].y, sh) + fmaf(rc, rv[u].y, rh), 0.f);
                o.z = fmaxf(fmaf(sc, av[u].z, sh) + fmaf(rc, rv[u].z, rh), 0.f);
                o.w = fmaxf(fmaf(sc, av[u].w, sh) + fmaf(rc, rv[u].w, rh), 0.f);
                *reinterpret_cast<float4*>(out + base + p) = o;
            }
        }
    } else {
        for (int p = p0 + tid; p < min(P, p0 + EW_TILE); p += 256)
            out[base + p] = fmaxf(fmaf(sc, a3[base + p], sh) + fmaf(rc, res[base + p], rh), 0.f);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_add_relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                              const float* __restrict__ a3, const float* __restrict__ ad,
                                                              float* __restrict__ g, float* __restrict__ partial,
                                                              float* __restrict__ partial_d, int P, int tiles) {
    __shared__ float red[4 * 3];
    const int row = blockIdx.x / tiles, tix = blockIdx.x - row * tiles;
    const size_t base = (size_t)row * P;
    const int p0 = tix * EW_TILE;
    float v[3] = {0.f, 0.f, 0.f};
    if (VEC) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + (u * 256 + threadIdx.x) * 4;
            if (p < P) {
                const float4 d = *reinterpret_cast<const float4*>(dout + base + p);
                const float4 o = *reinterpret_cast<const float4*>(out + base + p);
                const float4 a = *reinterpret_cast<const float4*>(a3 + base + p);
                float4 gg;
                gg.x = o.x > 0.f ? d.x : 0.f;
                gg.y = o.y > 0.f ? d.y : 0.f;
                gg.z = o.z > 0.f ? d.z : 0.f;
                gg.w = o.w > 0.f ? d.w : 0.f;
                *reinterpret_cast<float4*>(g + base + p) = gg;
                v[0] += (gg.x + gg.y) + (gg.z + gg.w);
                v[1] += fmaf(gg.x, a.x, fmaf(gg.y, a.y, fmaf(gg.z, a.z, gg.w * a.w)));
                if (ad != nullptr) {
                    const float4 b = *reinterpret_cast<const float4*>(ad + base + p);
                    v[2] += fmaf(gg.x, b.x, fmaf(gg.y, b.y, fmaf(gg.z, b.z, gg.w * b.w)));
                }
            }
        }
    } else {
        for (int p = p0 + threadIdx.x; p < min(P, p0 + EW_TILE); p += 256) {
            const float gg = out[base + p] > 0.f ? dout[base + p] : 0.f;
            g[base + p] = gg;
            v[0] += gg;
            v[1] = fmaf(gg, a3[base + p], v[1]);
            if (ad != nullptr) v[2] = fmaf(gg, ad[base + p], v[2]);
        }
    }
    float o3[3];
    block_sum_256<3>(v, red, o3);
    if (threadIdx.x == 0) {
        partial[((size_t)row * tiles + tix) * 2] = o3[0];
        partial[((size_t)row * tiles + tix) * 2 + 1] = o3[1];
        if (partial_d != nullptr) {
            partial_d[((size_t)row * tiles + tix) * 2] = o3[0];
            partial_d[((size_t)row * tiles + tix) * 2 + 1] = o3[2];
        }
    }
}

// segs = 1: AdaptiveAvgPool3d((1,1,1)) (task 'class', x3d.py:239); segs = T: AdaptiveAvgPool3d((None,1,1)) (task 'loc',
// x3d.py:241): segment s of a (n, c) row is the contiguous plane s (P / segs elements).  pooled is [N][C][segs].
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ a5, const float* __restrict__ c5,
                                                               float* __restrict__ pooled, int P, int segs) {
    __shared__ float red[4];
    const int row = blockIdx.x / segs, seg = blockIdx.x - row * segs;
    const int Ps = P / segs;
    const float sc = c5[(size_t)row * 2], sh = c5[(size_t)row * 2 + 1];
    const float* p = a5 + (size_t)row * P + (size_t)seg * Ps;
    float v[1] = {0.f};
    for (int i = threadIdx.x; i < Ps; i += 256) v[0] += fmaxf(fmaf(sc, p[i], sh), 0.f);
    float o[1];
    block_sum_256<1>(v, red, o);
    if (threadIdx.x == 0) pooled[blockIdx.x] = o[0] / (float)Ps;
}

__global__ __launch_bounds__(256) void bn_relu_pool_bwd_kernel(const float* __restrict__ a5, const float* __restrict__ c5,
                                                               const float* __restrict__ dpooled, float* __restrict__ g,
                                                               float* __restrict__ partial, int P, int tiles, int segs) {
    __shared__ float red[4 * 2];
    const int row = blockIdx.x / tiles, tix = blockIdx.x - row * tiles;
    const int Ps = P / segs;
    const float sc = c5[(size_t)row * 2], sh = c5[(size_t)row * 2 + 1];
    const float inv = 1.f / (float)Ps;
    const size_t base = (size_t)row * P;
    const int p0 = tix * EW_TILE;
    float v[2] = {0.f, 0.f};
    for (int p = p0 + threadIdx.x; p < min(P, p0 + EW_TILE); p += 256) {
        const float a = a5[base + p];
        const float d = dpooled[(size_t)row * segs + (segs == 1 ? 0 : p / Ps)] * inv;
        const float gg = fmaf(sc, a, sh) > 0.f ? d : 0.f;
        g[base + p] = gg;
        v[0] += gg;
        v[1] = fmaf(gg, a, v[1]);
    }
    float o[2];
    block_sum_256<2>(v, red, o);
    if (threadIdx.x == 0) {
        partial[((size_t)row * tiles + tix) * 2] = o[0];
        partial[((size_t)row * tiles + tix) * 2 + 1] = o[1];
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                  size_t n, float lr, float mu, float wd, float gs, int first) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float wi = w[i];
    const float gi = fmaf(wd, wi, g[i] * gs);
    const float mi = first ? gi : fmaf(mu, m[i], gi);
    m[i] = mi;
    w[i] = wi - lr * mi;
}

// Stand-alone SubBatchNorm3d (x3d.py:47-58 called as a module of its own): per-(sample, channel) row statistics and the
// per-row affine map.  g == NULL: partial = {sum x, sum x^2} (forward statistics); else {sum g, sum g*x} (backward).
__global__ __launch_bounds__(256) void bn_rowstats_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          float* __restrict__ partial, int P, int tiles) {
    __shared__ float red[4 * 2];
    const int row = blockIdx.x / tiles, tix = blockIdx.x - row * tiles;
    const size_t base = (size_t)row * P;
    const int p0 = tix * EW_TILE;
    float v[2] = {0.f, 0.f};
    for (int p = p0 + threadIdx.x; p < min(P, p0 + EW_TILE); p += 256) {
        const float xv = x[base + p];
        if (g == nullptr) { v[0] += xv; v[1] = fmaf(xv, xv, v[1]); }
        else { const float gv = g[base + p]; v[0] += gv; v[1] = fmaf(gv, xv, v[1]); }
    }
    float o[2];
    block_sum_256<2>(v, red, o);
    if (threadIdx.x == 0) {
        partial[((size_t)row * tiles + tix) * 2] = o[0];
        partial[((size_t)row * tiles + tix) * 2 + 1] = o[1];
    }
}

// ncoef 2: out = c0 * x + c1;  ncoef 3: out = c0 * g + c1 * x + c2   (coef per (sample, channel) row)
__global__ __launch_bounds__(256) void bn_affine_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ coef, float* __restrict__ out, int P,
                                                        int tiles, int ncoef) {
    const int row = blockIdx.x / tiles, tix = blockIdx.x - row * tiles;
    const size_t base = (size_t)row * P;
    const int p0 = tix * EW_TILE;
    const float c0 = coef[(size_t)row * ncoef], c1 = coef[(size_t)row * ncoef + 1];
    const float c2 = ncoef == 3 ? coef[(size_t)row * 3 + 2] : 0.f;
    for (int p = p0 + threadIdx.x; p < min(P, p0 + EW_TILE); p += 256)
        out[base + p] = ncoef == 3 ? fmaf(c0, g[base + p], fmaf(c1, x[base + p], c2)) : fmaf(c0, x[base + p], c1);
}

// acc = (first ? 0 : acc) + scale * g  (gradient accumulation over micro-batches: loss / num_steps_per_update,
// train_x3d_kinetics_multigrid.py:267-273)
__global__ __launch_bounds__(256) void grad_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ g, size_t n,
                                                              float scale, int first) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    acc[i] = first ? scale * g[i] : fmaf(scale, g[i], acc[i]);
}

}  // namespace

extern "C" int x3d_ew_tiles(int P) { return cdiv(P, EW_TILE); }

extern "C" size_t x3d_finalize_scratch_bytes(int N, int C, int Wd) {
    return (size_t)N * C * 2 * sizeof(double) + ((size_t)N * C * 2 + (size_t)N * Wd) * sizeof(float) + 64;
}

extern "C" int x3d_bn_fwd_finalize(const float* partial, int N, int C, int tiles, int S, int count,
                                   const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* coef, float* save,
                                   float* nsum, void* scratch, void* stream) {
    X3D_CHECK_ARG(partial && gamma && beta && coef && save && scratch);
    X3D_CHECK_ARG(N > 0 && C > 0 && tiles > 0 && S > 0 && count > 0);
    if (N % S != 0) {
        x3d_set_error("split BN needs batch %% num_splits == 0 (got N=%d, splits=%d; x3d.py:50)", N, S);
        return X3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    if (S <= BN_MAXS && N <= BN_MAXN) {
        hipLaunchKernelGGL(bn_fwd_fused_kernel, dim3(C), dim3(256), 0, s, partial, N, C, tiles, S, count, gamma, beta,
                           running_mean, running_var, momentum, eps, coef, save, nsum);
    } else {
        double* dsum = (double*)scratch;
        hipLaunchKernelGGL(reduce_tiles_kernel<2>, dim3(cdiv(N * C, 4)), dim3(256), 0, s, partial, dsum, N * C, tiles);
        hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, dsum, N, C, S, count, gamma,
                           beta, running_mean, running_var, momentum, eps, coef, save, nsum);
    }
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_eval_coef(const float* running_mean, const float* running_var, const float* gamma,
                                const float* beta, float eps, int N, int C, float* coef, void* stream) {
    X3D_CHECK_ARG(running_mean && running_var && gamma && beta && coef && N > 0 && C > 0);
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       running_mean, running_var, gamma, beta, eps, N, C, coef);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_se_fwd(const float* coef, const float* nsum, int N, int C, int Wd, int count, const float* w1,
                          const float* b1, const float* w2, const float* b2, float* coef_out, float* save_se,
                          float* save_z, float* save_pool, void* stream) {
    X3D_CHECK_ARG(coef && nsum && w1 && b1 && w2 && b2 && coef_out && save_se && save_z && save_pool);
    X3D_CHECK_ARG(N > 0 && C > 0 && C <= SE_MAXC && Wd > 0 && Wd <= SE_MAXW && count > 0);
    hipLaunchKernelGGL(se_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, coef, nsum, C, Wd, count, w1, b1,
                       w2, b2, coef_out, save_se, save_z, save_pool);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

template <int UPW>
static void se_bn_fwd_launch(const SeBnArgs& A, hipStream_t s) {
    const int cpl = cdiv(A.C, 64);
    if (cpl <= 2) hipLaunchKernelGGL((se_bn_fwd_kernel<UPW, 2>), dim3(A.N), dim3(1024), 0, s, A);
    else if (cpl <= 4) hipLaunchKernelGGL((se_bn_fwd_kernel<UPW, 4>), dim3(A.N), dim3(1024), 0, s, A);
    else if (cpl <= 8) hipLaunchKernelGGL((se_bn_fwd_kernel<UPW, 8>), dim3(A.N), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL((se_bn_fwd_kernel<UPW, 16>), dim3(A.N), dim3(1024), 0, s, A);
}

extern "C" int x3d_se_bn_fwd(const float* partial, int N, int C, int tiles, int S, int count, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                             int Wd, const float* w1, const float* b1, const float* w2, const float* b2,
                             float* coef_out, float* save, float* nsum, float* save_se, float* save_z,
                             float* save_pool, void* stream) {
    X3D_CHECK_ARG(partial && gamma && beta && w1 && b1 && w2 && b2 && coef_out && save && nsum && save_se && save_z && save_pool);
    X3D_CHECK_ARG(N > 0 && C > 0 && C <= SE_MAXC && Wd > 0 && Wd <= SE_MAXW && tiles > 0 && S > 0 && count > 0);
    if (N % S != 0) {
        x3d_set_error("split BN needs batch %% num_splits == 0 (got N=%d, splits=%d; x3d.py:50)", N, S);
        return X3D_EINVAL;
    }
    SeBnArgs A;
    A.partial = partial; A.gamma = gamma; A.beta = beta; A.rmean = running_mean; A.rvar = running_var;
    A.w1 = w1; A.b1 = b1; A.w2 = w2; A.b2 = b2; A.coef_out = coef_out; A.save = save; A.nsum = nsum;
    A.save_se = save_se; A.save_z = save_z; A.save_pool = save_pool;
    A.N = N; A.C = C; A.tiles = tiles; A.S = S; A.count = count; A.Wd = Wd; A.momentum = momentum; A.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if (Wd <= 16) se_bn_fwd_launch<1>(A, s);
    else if (Wd <= 32) se_bn_fwd_launch<2>(A, s);
    else se_bn_fwd_launch<4>(A, s);
    x3d_note_kernel("se_bn_fwd_kernel");
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_bwd_finalize(const float* partial, int N, int C, int tiles, int S, int count,
                                   const float* gamma, const float* save, float* cb, float* dgamma, float* dbeta,
                                   int accumulate, void* scratch, void* stream) {
    X3D_CHECK_ARG(partial && gamma && save && cb && dgamma && dbeta && scratch);
    X3D_CHECK_ARG(N > 0 && C > 0 && tiles > 0 && S > 0 && count > 0 && N % S == 0);
    hipStream_t s = (hipStream_t)stream;
    if (S <= BN_MAXS && N <= BN_MAXN) {
        hipLaunchKernelGGL(bn_bwd_fused_kernel, dim3(C), dim3(256), 0, s, partial, N, C, tiles, S, count, gamma, save,
                           cb, dgamma, dbeta, accumulate);
    } else {
        double* dsum = (double*)scratch;
        hipLaunchKernelGGL(reduce_tiles_kernel<2>, dim3(cdiv(N * C, 4)), dim3(256), 0, s, partial, dsum, N * C, tiles);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, dsum, N, C, S, count, gamma,
                           save, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, cb, dgamma,
                           dbeta, accumulate);
    }
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_se_bn_bwd_finalize(const float* partial, int N, int C, int tiles, int S, int count,
                                      const float* gamma, const float* beta, const float* save, const float* nsum,
                                      int Wd, const float* w1, const float* w2, const float* save_se,
                                      const float* save_z, const float* save_pool, float* cb, float* dgamma,
                                      float* dbeta, float* dw1, float* db1, float* dw2, float* db2, void* scratch,
                                      void* stream) {
    X3D_CHECK_ARG(partial && gamma && beta && save && nsum && w1 && w2 && save_se && save_z && save_pool);
    X3D_CHECK_ARG(cb && dgamma && dbeta && dw1 && db1 && dw2 && db2 && scratch);
    X3D_CHECK_ARG(N > 0 && C > 0 && C <= SE_MAXC && Wd > 0 && Wd <= SE_MAXW && tiles > 0 && S > 0 && N % S == 0);
    hipStream_t s = (hipStream_t)stream;
    double* dsum = (double*)scratch;
    float* dpool = (float*)(dsum + (size_t)N * C * 2);
    float* dz2 = dpool + (size_t)N * C;
    float* dz1 = dz2 + (size_t)N * C;
    // threads per row of the merged tile-reduction + per-sample kernel: the largest power of two with C * tpr <= 1024
    int tpr = 1;
    while (tpr < 16 && C * tpr * 2 <= 1024) tpr *= 2;
    // (round 4) one launch instead of two where a thread sums at most 32 tile pairs (stages 3-4 of every model at the
    // multigrid shapes: 98 / 25 tiles per row; the stage 1-2 rows have 392-1568 tiles and keep the all-CU reduction)
    if (!x3d_opt(X3D_OPT_NO_SE_BWD_MERGE) && cdiv(tiles, tpr) <= 32) {
        SeBwdArgs B;
        B.partial = partial; B.dsum = dsum; B.gamma = gamma; B.beta = beta; B.save = save; B.w1 = w1; B.w2 = w2;
        B.save_se = save_se; B.save_z = save_z; B.dpool = dpool; B.dz2o = dz2; B.dz1o = dz1;
        B.C = C; B.S = S; B.Wd = Wd; B.tiles = tiles; B.tpr = tpr;
        if (Wd <= 16) hipLaunchKernelGGL((se_bwd_sample2_kernel<1>), dim3(N), dim3(1024), 0, s, B);
        else if (Wd <= 32) hipLaunchKernelGGL((se_bwd_sample2_kernel<2>), dim3(N), dim3(1024), 0, s, B);
        else hipLaunchKernelGGL((se_bwd_sample2_kernel<4>), dim3(N), dim3(1024), 0, s, B);
    } else {
        hipLaunchKernelGGL(reduce_tiles_kernel<2>, dim3(cdiv(N * C, 4)), dim3(256), 0, s, partial, dsum, N * C, tiles);
        hipLaunchKernelGGL(se_bwd_sample_kernel, dim3(N), dim3(256), 0, s, dsum, C, S, Wd, gamma, beta, save, w1, w2,
                           save_se, save_z, dpool, dz2, dz1);
    }
    const int nbw = cdiv(C * Wd, 256);
    hipLaunchKernelGGL(se_tail_kernel, dim3(nbw + (64 % S == 0 ? cdiv(C, 4) : cdiv(C, 256))), dim3(256), 0, s, nbw, dsum, N, C, S, count, Wd, gamma, save,
                       save_se, dpool, nsum, cb, dgamma, dbeta, dz2, dz1, save_z, save_pool, dw1, db1, dw2, db2);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_add_relu_fwd(const float* a3, const float* c3, const float* res, const float* cd, float* out,
                                   int N, int C, int P, void* stream) {
    X3D_CHECK_ARG(a3 && c3 && res && out && N > 0 && C > 0 && P > 0);
    const int ewt = cdiv(P, EW_TILE);
    X3D_CHECK_ARG((long long)N * C * ewt <= 0x7fffffffLL);
    dim3 grid((unsigned)(N * C * ewt)), block(256);
    if (P % 4 == 0)
        hipLaunchKernelGGL(bn_add_relu_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, a3, c3, res, cd, out, P, ewt);
    else
        hipLaunchKernelGGL(bn_add_relu_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, a3, c3, res, cd, out, P, ewt);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_stats_add_relu_fwd(const float* a3, const float* partial, int tiles, int S, int count,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         float momentum, float eps, float* save, const float* res, const float* cd,
                                         float* out, int N, int C, int P, void* stream) {
    X3D_CHECK_ARG(a3 && partial && gamma && beta && save && res && out && N > 0 && C > 0 && P > 0 && tiles > 0);
    const int ewt = cdiv(P, EW_TILE);
    X3D_CHECK_ARG(S > 0 && N % S == 0 && count > 0 && (long long)N * C * ewt <= 0x7fffffffLL);
    X3D_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
    dim3 grid((unsigned)(N * C * ewt)), block(256);
    if (P % 4 == 0)
        hipLaunchKernelGGL(bn_stats_add_relu_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, a3, partial, tiles, N, C,
                           S, count, gamma, beta, running_mean, running_var, momentum, eps, save, res, cd, out, P, ewt);
    else
        hipLaunchKernelGGL(bn_stats_add_relu_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, a3, partial, tiles, N, C,
                           S, count, gamma, beta, running_mean, running_var, momentum, eps, save, res, cd, out, P, ewt);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_add_relu_bwd(const float* dout, const float* out, const float* a3, const float* ad, float* g,
                                   float* partial, float* partial_d, int N, int C, int P, void* stream) {
    X3D_CHECK_ARG(dout && out && a3 && g && partial && N > 0 && C > 0 && P > 0);
    X3D_CHECK_ARG((ad == nullptr) == (partial_d == nullptr));
    const int tiles = cdiv(P, EW_TILE);
    X3D_CHECK_ARG((long long)N * C * tiles <= 0x7fffffffLL);
    dim3 grid((unsigned)(N * C * tiles)), block(256);
    if (P % 4 == 0)
        hipLaunchKernelGGL(bn_add_relu_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, dout, out, a3, ad, g,
                           partial, partial_d, P, tiles);
    else
        hipLaunchKernelGGL(bn_add_relu_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, dout, out, a3, ad, g,
                           partial, partial_d, P, tiles);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_relu_pool_fwd(const float* a5, const float* c5, float* pooled, int N, int C, int P, int segs,
                                    void* stream) {
    X3D_CHECK_ARG(a5 && c5 && pooled && N > 0 && C > 0 && P > 0 && segs > 0 && P % segs == 0);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(N * C * segs), dim3(256), 0, (hipStream_t)stream, a5, c5, pooled, P,
                       segs);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_relu_pool_bwd(const float* a5, const float* c5, const float* dpooled, float* g, float* partial,
                                    int N, int C, int P, int segs, void* stream) {
    X3D_CHECK_ARG(a5 && c5 && dpooled && g && partial && N > 0 && C > 0 && P > 0);
    X3D_CHECK_ARG(segs > 0 && P % segs == 0);
    const int tiles = cdiv(P, EW_TILE);
    X3D_CHECK_ARG((long long)N * C * tiles <= 0x7fffffffLL);
    hipLaunchKernelGGL(bn_relu_pool_bwd_kernel, dim3((unsigned)(N * C * tiles)), dim3(256), 0, (hipStream_t)stream, a5, c5, dpooled,
                       g, partial, P, tiles, segs);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_rowstats(const float* x, const float* g, float* partial, int N, int C, int P, void* stream) {
    X3D_CHECK_ARG(x && partial && N > 0 && C > 0 && P > 0);
    const int tiles = cdiv(P, EW_TILE);
    X3D_CHECK_ARG((long long)N * C * tiles <= 0x7fffffffLL);
    hipLaunchKernelGGL(bn_rowstats_kernel, dim3((unsigned)(N * C * tiles)), dim3(256), 0, (hipStream_t)stream, x, g, partial, P,
                       tiles);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_bn_affine(const float* x, const float* g, const float* coef, float* out, int N, int C, int P, int ncoef,
                             void* stream) {
    X3D_CHECK_ARG(x && coef && out && N > 0 && C > 0 && P > 0 && (ncoef == 2 || (ncoef == 3 && g != nullptr)));
    const int tiles = cdiv(P, EW_TILE);
    X3D_CHECK_ARG((long long)N * C * tiles <= 0x7fffffffLL);
    hipLaunchKernelGGL(bn_affine_kernel, dim3((unsigned)(N * C * tiles)), dim3(256), 0, (hipStream_t)stream, x, g, coef, out, P,
                       tiles, ncoef);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_grad_accumulate(float* acc, const float* g, size_t n, float scale, int first, void* stream) {
    X3D_CHECK_ARG(acc && g && n > 0);
    hipLaunchKernelGGL(grad_accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc, g, n,
                       scale, first);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_sgd_fused(float* w, const float* g, float* m, size_t n, float lr, float momentum,
                             float weight_decay, float grad_scale, int first, void* stream) {
    X3D_CHECK_ARG(w && g && m && n > 0);
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, g, m, n, lr,
                       momentum, weight_decay, grad_scale, first);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
