// Classification head (x3d.py:333-343): fc1 (1x1x1 conv on the pooled vector, no bias) -> ReLU -> Dropout ->
// fc2 (Linear + bias), the cross-entropy loss of the training script (train_x3d_kinetics_multigrid.py:189,259:
// nn.CrossEntropyLoss, mean over rows) and their backward.  R rows = N (task 'class') or N*T (task 'loc', per-frame
// logits, x3d.py:340-343); J = 2048 hidden features; C classes; K = pooled channels (432 / 630).
//
// Everything here is a few MB of weights against a handful of rows: each kernel streams its weight matrix once with
// coalesced loads and keeps the rows in registers / LDS, eight rows (HB) at a time; fp32 VALU FMAs throughout (exact
// fp32, no split).  Sums run in a fixed order (no atomics): bitwise reproducible.
//   forward : head_fc1_kernel (+ReLU + dropout), head_fc2_kernel (+bias), head_ce_kernel (per-row loss + dlogits),
//             head_mean_kernel (loss = mean over rows; advances the dropout counter)
//   backward: head_bwd_fc2_kernel (dW2, db2, class-slice partials of d hidden),
//             head_bwd_fc1_kernel (dW1 | feature-slice partials of d pooled), head_sum_slices_kernel (d pooled)
// Dropout: counter-based hash of (seed, draw counter, element index); the draw counter lives in device memory and is
// advanced by the LAST forward kernel, so a replayed hipGraph draws a fresh mask every step.
#include "common.h"

namespace {

constexpr int HB = 8;            // rows per register block
constexpr int H_CS = 16;         // class slices of the fc2 backward
constexpr int H_JS = 64;         // hidden-feature slices of the d-pooled reduction

__device__ __forceinline__ unsigned hash_u32(unsigned long long seed, unsigned long long ctr, unsigned idx) {
    // splitmix64 finaliser over (seed, counter, index): uniform 32 bits per element, no state
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1) + ((unsigned long long)idx << 1 | 1ull) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned)(z >> 32);
}

// hd[r][j] = dropout(relu(sum_k W1[j][k] * pooled[r][k])): one wave per hidden feature, lanes along k
__global__ __launch_bounds__(256) void head_fc1_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                       float* __restrict__ hd, int R, int K, int J, float p_drop,
                                                       const unsigned long long* __restrict__ rng) {
    extern __shared__ float prow[];                        // [HB][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    const unsigned long long seed = rng ? rng[0] : 0ull, ctr = rng ? rng[1] : 0ull;
    const unsigned thr = p_drop > 0.f ? (unsigned)fminf(p_drop * 4294967296.f, 4294967295.f) : 0u;
    const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int r0 = 0; r0 < R; r0 += HB) {
        __syncthreads();
        for (int i = threadIdx.x; i < HB * K; i += 256) {
            const int b = i / K, k = i - b * K;
            prow[i] = (r0 + b < R) ? pooled[(size_t)(r0 + b) * K + k] : 0.f;
        }
        __syncthreads();
        if (j < J) {
            float acc[HB];
#pragma unroll
            for (int b = 0; b < HB; ++b) acc[b] = 0.f;
            const float* wr = w1 + (size_t)j * K;
            for (int k = lane; k < K; k += 64) {
                const float w = wr[k];
#pragma unroll
                for (int b = 0; b < HB; ++b) acc[b] = fmaf(w, prow[b * K + k], acc[b]);
            }
#pragma unroll
            for (int b = 0; b < HB; ++b) acc[b] = wave_sum(acc[b]);
            if (lane < HB && r0 + lane < R) {
                float v = 0.f;
#pragma unroll
                for (int b = 0; b < HB; ++b) v = lane == b ? acc[b] : v;
                v = fmaxf(v, 0.f);
                if (p_drop > 0.f) {
                    const unsigned u = hash_u32(seed, ctr, (unsigned)((r0 + lane) * J + j));
                    v = u >= thr ? v * keep_scale : 0.f;
                }
                hd[(size_t)(r0 + lane) * J + j] = v;
            }
        }
    }
}

// logits[r][c] = b2[c] + sum_j W2[c][j] * hd[r][j]: one wave per class, lanes along j (float4)
__global__ __launch_bounds__(256) void head_fc2_kernel(const float* __restrict__ hd, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ logits, int R,
                                                       int J, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    const float* wr = w2 + (size_t)c * J;
    for (int r0 = 0; r0 < R; r0 += HB) {
        float acc[HB];
#pragma unroll
        for (int b = 0; b < HB; ++b) acc[b] = 0.f;
        for (int j4 = lane * 4; j4 < J; j4 += 256) {
            const float4 w = *reinterpret_cast<const float4*>(wr + j4);
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                const int r = min(r0 + b, R - 1);
                const float4 h = *reinterpret_cast<const float4*>(hd + (size_t)r * J + j4);
                acc[b] = fmaf(w.x, h.x, fmaf(w.y, h.y, fmaf(w.z, h.z, fmaf(w.w, h.w, acc[b]))));
            }
        }
#pragma unroll
        for (int b = 0; b < HB; ++b) acc[b] = wave_sum(acc[b]);
        if (lane < HB && r0 + lane < R) {
            float v = 0.f;
#pragma unroll
            for (int b = 0; b < HB; ++b) v = lane == b ? acc[b] : v;
            logits[(size_t)(r0 + lane) * C + c] = v + b2[c];
        }
    }
}

// per row: loss_r = logsumexp(logits[r]) - logits[r][label]; dlogits[r][c] = (softmax - onehot) / R   (mean reduction)
__global__ __launch_bounds__(256) void head_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                      float* __restrict__ loss_rows, float* __restrict__ dlogits, int R,
                                                      int C) {
    __shared__ float red[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (size_t)r * C;
    float m = -__builtin_inff();
    for (int c = tid; c < C; c += 256) m = fmaxf(m, lr[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < C; c += 256) s += expf(lr[c] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const int lab = (int)labels[r];
    const float lse = m + logf(s);
    if (tid == 0) loss_rows[r] = lse - lr[lab];
    const float invR = 1.f / (float)R, invs = 1.f / s;
    for (int c = tid; c < C; c += 256) dlogits[(size_t)r * C + c] = (expf(lr[c] - m) * invs - (c == lab ? 1.f : 0.f)) * invR;
}

// loss = mean(loss_rows); the last kernel of the head's forward: advances the dropout draw counter
__global__ __launch_bounds__(256) void head_mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out,
                                                        unsigned long long* __restrict__ rng) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
        if (rng != nullptr) rng[1] += 1ull;
    }
}

// fc2 backward.  grid (ceil(J / 256), H_CS): thread = hidden feature j, block slice = classes [c0, c1).
//   dW2[c][j] = sum_r dlog[r][c] * hd[r][j]          (written here, complete: the r loop is inside)
//   db2[c]    = sum_r dlog[r][c]                      (blockIdx.x == 0)
//   dpart[s][r][j] = sum_{c in slice s} dlog[r][c] * W2[c][j]
__global__ __launch_bounds__(256) void head_bwd_fc2_kernel(const float* __restrict__ dlog, const float* __restrict__ hd,
                                                           const float* __restrict__ w2, float* __restrict__ dw2,
                                                           float* __restrict__ db2, float* __restrict__ dpart, int R, int J,
                                                           int C) {
    extern __shared__ float dl[];                          // [HB][cs] dlogits of the row block and class slice
    const int j = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    const int cs = cdiv(C, H_CS), c0 = s * cs, c1 = min(C, c0 + cs);
    const bool jv = j < J;
    for (int r0 = 0; r0 < R; r0 += HB) {
        __syncthreads();
        for (int i = threadIdx.x; i < HB * cs; i += 256) {
            const int b = i / cs, c = c0 + i - b * cs;
            dl[i] = (r0 + b < R && c < c1) ? dlog[(size_t)(r0 + b) * C + c] : 0.f;
        }
        __syncthreads();
        float hv[HB], part[HB];
#pragma unroll
        for (int b = 0; b < HB; ++b) { hv[b] = (jv && r0 + b < R) ? hd[(size_t)(r0 + b) * J + j] : 0.f; part[b] = 0.f; }
        for (int c = c0; c < c1; ++c) {
            const float w = jv ? w2[(size_t)c * J + j] : 0.f;
            float dw = 0.f;
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                const float d = dl[b * cs + (c - c0)];
                part[b] = fmaf(d, w, part[b]);
                dw = fmaf(d, hv[b], dw);
            }
            if (jv) {
                float* pw = dw2 + (size_t)c * J + j;
                *pw = (r0 == 0) ? dw : *pw + dw;           // later row blocks accumulate in order
            }
        }
        if (jv) {
#pragma unroll
            for (int b = 0; b < HB; ++b)
                if (r0 + b < R) dpart[((size_t)s * R + r0 + b) * J + j] = part[b];
        }
    }
    if (blockIdx.x == 0) {
        for (int c = c0 + threadIdx.x; c < c1; c += 256) {
            float sb = 0.f;
            for (int r = 0; r < R; ++r) sb += dlog[(size_t)r * C + c];
            db2[c] = sb;
        }
    }
}

// dh[r][j] = hd[r][j] > 0 ? keep_scale * sum_s dpart[s][r][j] : 0   (ReLU and dropout backward: hd > 0 <=> kept and positive)
__device__ __forceinline__ float head_dh(const float* __restrict__ dpart, const float* __restrict__ hd, int R, int J, int r,
                                         int j, float keep_scale) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < H_CS; ++k) s += dpart[((size_t)k * R + r) * J + j];
    return hd[(size_t)r * J + j] > 0.f ? s * keep_scale : 0.f;
}

// fc1 backward, two roles in one launch:
//   blocks [0, J / 4): one wave per hidden feature j:  dW1[j][k] = sum_r dh[r][j] * pooled[r][k]   (lanes along k)
//   blocks [J / 4, J / 4 + H_JS * kb): thread = pooled channel k, slice of features:
//                                    ppart[s][r][k] = sum_{j in slice} dh[r][j] * W1[j][k]
__global__ __launch_bounds__(256) void head_bwd_fc1_kernel(const float* __restrict__ dpart, const float* __restrict__ hd,
                                                           const float* __restrict__ pooled, const float* __restrict__ w1,
                                                           float* __restrict__ dw1, float* __restrict__ ppart, int R, int K,
                                                           int J, float keep_scale, int nb1, int kb) {
    if ((int)blockIdx.x < nb1) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int j = blockIdx.x * 4 + wave;
        if (j >= J) return;
        float acc[10];                                      // K <= 640: ten 64-channel strips
#pragma unroll
        for (int ki = 0; ki < 10; ++ki) acc[ki] = 0.f;
        for (int r0 = 0; r0 < R; r0 += 64) {
            const float dhv = (r0 + lane < R) ? head_dh(dpart, hd, R, J, r0 + lane, j, keep_scale) : 0.f;   // lane = row
            const int nr = min(64, R - r0);
            for (int rr = 0; rr < nr; ++rr) {
                const float d = __shfl(dhv, rr);
                const float* pr = pooled + (size_t)(r0 + rr) * K;
#pragma unroll
                for (int ki = 0; ki < 10; ++ki) {
                    const int k = lane + 64 * ki;
                    acc[ki] = fmaf(d, pr[k < K ? k : 0], acc[ki]);
                }
            }
        }
#pragma unroll
        for (int ki = 0; ki < 10; ++ki) {
            const int k = lane + 64 * ki;
            if (k < K) dw1[(size_t)j * K + k] = acc[ki];
        }
        return;
    }
    const int bi = blockIdx.x - nb1;
    const int s = bi / kb, k = (bi - s * kb) * 256 + threadIdx.x;
    const int js = cdiv(J, H_JS), j0 = s * js, j1 = min(J, j0 + js);
    __shared__ float dhs[HB][128];                          // dh of the row block for a run of 128 features
    for (int r0 = 0; r0 < R; r0 += HB) {
        float acc[HB];
#pragma unroll
        for (int b = 0; b < HB; ++b) acc[b] = 0.f;
        for (int jj = j0; jj < j1; jj += 128) {
            __syncthreads();
            for (int i = threadIdx.x; i < HB * 128; i += 256) {
                const int b = i >> 7, j = jj + (i & 127);
                dhs[b][i & 127] = (r0 + b < R && j < j1) ? head_dh(dpart, hd, R, J, r0 + b, j, keep_scale) : 0.f;
            }
            __syncthreads();
            if (k < K) {
                const int je = min(128, j1 - jj);
                for (int t = 0; t < je; ++t) {
                    const float w = w1[(size_t)(jj + t) * K + k];
#pragma unroll
                    for (int b = 0; b < HB; ++b) acc[b] = fmaf(dhs[b][t], w, acc[b]);
                }
            }
        }
        if (k < K) {
#pragma unroll
            for (int b = 0; b < HB; ++b)
                if (r0 + b < R) ppart[((size_t)s * R + r0 + b) * K + k] = acc[b];
        }
    }
}

// out[i] = sum_s part[s][i]
__global__ __launch_bounds__(256) void head_sum_slices_kernel(const float* __restrict__ part, float* __restrict__ out, int n,
                                                              int slices) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // four independent partial sums (fixed order): the loads of a round are in flight together
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < slices; k += 4) {
        s0 += part[(size_t)k * n + i]; s1 += part[(size_t)(k + 1) * n + i];
        s2 += part[(size_t)(k + 2) * n + i]; s3 += part[(size_t)(k + 3) * n + i];
    }
    for (; k < slices; ++k) s0 += part[(size_t)k * n + i];
    out[i] = (s0 + s1) + (s2 + s3);
}

// Charades localisation losses (train_x3d_charades_loc.py:123,168-189): per-frame logits [B][C][T] are linearly interpolated
// to the label length TL (F.interpolate(mode='linear'), align_corners False), then
//   loc_loss = BCEWithLogits(interp, labels)                      (mean over B*C*TL)
//   cls_loss = BCEWithLogits(max_t interp, max_t labels)          (mean over B*C)
//   loss     = (cls_loss + loc_loss) / (2 * num_steps_per_update)
// One thread per (b, c) row: forward sums and the gradient w.r.t. the un-interpolated logits (scatter of two taps per frame).
__device__ __forceinline__ float bce_logits(float z, float y) { return fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z))); }

__global__ __launch_bounds__(256) void loc_loss_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                       float* __restrict__ dlogits, float* __restrict__ rowsum, int rows, int T,
                                                       int TL, float gscale) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float* z = logits + (size_t)row * T;
    const float* y = labels + (size_t)row * TL;
    float* dz = dlogits + (size_t)row * T;
    for (int t = 0; t < T; ++t) dz[t] = 0.f;
    const float scale = (float)T / (float)TL;
    const float gl = gscale / ((float)rows * (float)TL), gc = gscale / (float)rows;
    float loc = 0.f, zmax = -__builtin_inff(), ymax = -__builtin_inff();
    int imax0 = 0, imax1 = 0;
    float lmax = 0.f;
    for (int i = 0; i < TL; ++i) {
        float src = scale * ((float)i + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
        const int i0 = min((int)src, T - 1), i1 = i0 + (i0 < T - 1 ? 1 : 0);
        const float l1 = src - (float)i0, l0 = 1.f - l1;
        const float zi = l0 * z[i0] + l1 * z[i1];
        loc += bce_logits(zi, y[i]);
        const float d = (1.f / (1.f + expf(-zi)) - y[i]) * gl;
        dz[i0] += l0 * d;
        dz[i1] += l1 * d;
        if (zi > zmax) { zmax = zi; imax0 = i0; imax1 = i1; lmax = l1; }
        ymax = fmaxf(ymax, y[i]);
    }
    const float dc = (1.f / (1.f + expf(-zmax)) - ymax) * gc;
    dz[imax0] += (1.f - lmax) * dc;
    dz[imax1] += lmax * dc;
    rowsum[(size_t)row * 2] = bce_logits(zmax, ymax);
    rowsum[(size_t)row * 2 + 1] = loc;
}

// out[0] = mean over rows of rowsum[.][0]; out[1] = sum of rowsum[.][1] / (rows * TL)
__global__ __launch_bounds__(256) void loc_loss_reduce_kernel(const float* __restrict__ rowsum, int rows, int TL,
                                                              float* __restrict__ out) {
    __shared__ double red[4][2];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < rows; i += 256) { s0 += (double)rowsum[(size_t)i * 2]; s1 += (double)rowsum[(size_t)i * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s0; red[threadIdx.x >> 6][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (float)(((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) / (double)rows);
        out[1] = (float)(((red[0][1] + red[1][1]) + (red[2][1] + red[3][1])) / ((double)rows * (double)TL));
    }
}

}  // namespace

// losses[0] = cls_loss, losses[1] = loc_loss; dlogits = d[(cls + loc) * grad_scale] / d logits (the script uses
// grad_scale = 1 / (2 * num_steps_per_update)); scratch: 2 * B * C floats.
extern "C" int x3d_loc_losses(const float* logits, const float* labels, float* losses, float* dlogits, float* scratch, int B,
                              int C, int T, int TL, float grad_scale, void* stream) {
    X3D_CHECK_ARG(logits && labels && losses && dlogits && scratch && B > 0 && C > 0 && T > 0 && TL > 0);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * C;
    hipLaunchKernelGGL(loc_loss_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, logits, labels, dlogits, scratch, rows, T, TL,
                       grad_scale);
    hipLaunchKernelGGL(loc_loss_reduce_kernel, dim3(1), dim3(256), 0, s, scratch, rows, TL, losses);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" size_t x3d_head_scratch_floats(int R, int K, int J, int C) {
    (void)C;
    return (size_t)H_CS * R * J + (size_t)H_JS * R * K + (size_t)R;     // dpart, ppart, loss rows
}

// logits[R][C] from pooled[R][K]; hd[R][J] is kept for the backward.  rng = device {seed, draw counter} or NULL (p_drop 0).
extern "C" int x3d_head_fwd(const float* pooled, const float* w1, const float* w2, const float* b2, float* hd, float* logits,
                            int R, int K, int J, int C, float p_drop, const unsigned long long* rng, void* stream) {
    X3D_CHECK_ARG(pooled && w1 && w2 && b2 && hd && logits && R > 0 && K > 0 && J > 0 && C > 0);
    X3D_CHECK_ARG(J % 4 == 0 && p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng != nullptr));
    X3D_CHECK_ARG(K <= 640);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(head_fc1_kernel, dim3(cdiv(J, 4)), dim3(256), (size_t)HB * K * sizeof(float), s, pooled, w1, hd, R, K, J,
                       p_drop, rng);
    hipLaunchKernelGGL(head_fc2_kernel, dim3(cdiv(C, 4)), dim3(256), 0, s, hd, w2, b2, logits, R, J, C);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

// loss = mean_r CE(logits[r], labels[r]); dlogits = d loss / d logits.  Advances the dropout draw counter (rng may be NULL).
extern "C" int x3d_head_ce(const float* logits, const long long* labels, float* loss, float* dlogits, float* scratch, int R,
                           int C, unsigned long long* rng, void* stream) {
    X3D_CHECK_ARG(logits && labels && loss && dlogits && scratch && R > 0 && C > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(head_ce_kernel, dim3(R), dim3(256), 0, s, logits, labels, scratch, dlogits, R, C);
    hipLaunchKernelGGL(head_mean_kernel, dim3(1), dim3(256), 0, s, scratch, R, loss, rng);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

// advances the dropout draw counter without a loss (forward-only use of x3d_head_fwd in training mode)
extern "C" int x3d_head_advance_rng(unsigned long long* rng, float* dummy1, void* stream) {
    X3D_CHECK_ARG(rng && dummy1);
    hipLaunchKernelGGL(head_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dummy1, 1, dummy1, rng);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

// gradients of the head given dlogits[R][C]: dW1[J][K], dW2[C][J], db2[C], dpooled[R][K].
// scratch: x3d_head_scratch_floats(R, K, J, C) floats.
extern "C" int x3d_head_bwd(const float* dlogits, const float* hd, const float* pooled, const float* w1, const float* w2,
                            float* dw1, float* dw2, float* db2, float* dpooled, float* scratch, int R, int K, int J, int C,
                            float p_drop, void* stream) {
    X3D_CHECK_ARG(dlogits && hd && pooled && w1 && w2 && dw1 && dw2 && db2 && dpooled && scratch);
    X3D_CHECK_ARG(R > 0 && K > 0 && J > 0 && C > 0 && p_drop >= 0.f && p_drop < 1.f);
    hipStream_t s = (hipStream_t)stream;
    float* dpart = scratch;
    float* ppart = scratch + (size_t)H_CS * R * J;
    const int cs = cdiv(C, H_CS);
    hipLaunchKernelGGL(head_bwd_fc2_kernel, dim3(cdiv(J, 256), H_CS), dim3(256), (size_t)HB * cs * sizeof(float), s, dlogits,
                       hd, w2, dw2, db2, dpart, R, J, C);
    const int nb1 = cdiv(J, 4), kb = cdiv(K, 256);
    const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    hipLaunchKernelGGL(head_bwd_fc1_kernel, dim3(nb1 + H_JS * kb), dim3(256), 0, s, dpart, hd, pooled, w1, dw1, ppart, R, K, J,
                       keep_scale, nb1, kb);
    hipLaunchKernelGGL(head_sum_slices_kernel, dim3(cdiv(R * K, 256)), dim3(256), 0, s, ppart, dpooled, R * K, H_JS);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
