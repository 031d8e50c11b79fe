// Pointwise (1x1x1) convolution kernels: fp32 MFMA GEMMs with fused prologues/epilogues.
//
// Reference call sites replaced: conv1x1x1 (x3d.py:98-103) used as Bottleneck.conv1/conv3
// (:112,116,146,162), downsample[0] (:272), conv5 (:231,327) and their autograd backward.
//
// Mapping (all three kernels): v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, the same
// rounding as a VALU fmaf loop) with output channels on the MFMA row index and voxels on
// the column index, so every lane owns 4 consecutive voxels of a channel row:
//   - the B operand (activations, [C][P] row-major, P contiguous) is loaded straight from
//     HBM as one float4 per lane: 16 lanes x 16 B = 256 B contiguous per channel row, four
//     channel rows per wave instruction, no LDS round trip and no redundant loads;
//   - the A operand (weights, tiny) is staged per 32-channel chunk in LDS as [m][34]
//     (conflict-free ds_read_b32 for the 16x4 fragment);
//   - the result is stored as float4 per lane (256 B contiguous per channel row).
// BN statistics / BN-backward reductions ride in the epilogue: 16-lane DPP row sums, then
// one partial per (sample, channel, voxel tile) -- summed later in fixed order (fp64).
#include "common.h"

namespace {

enum { IN_RAW = 0, IN_AFFACT = 1, IN_BNBWD = 2 };
enum { EPI_STATS = 0, EPI_PLAIN = 1, EPI_ACTBWD = 2 };

constexpr int PW_BN = 256;    // voxels per workgroup tile (4 waves x 64)
constexpr int PW_KC = 32;     // channels per LDS weight chunk
constexpr int PW_KPAD = 34;   // LDS row stride of the weight chunk

struct PwArgs {
    const float* x;       // IN_RAW / IN_AFFACT: input [N][K][Pin];  IN_BNBWD: upstream grad g [N][K][P]
    const float* a;       // IN_BNBWD: raw forward output [N][K][P]
    const float* cin;     // IN_AFFACT: [N][K][2];  IN_BNBWD: [N][K][3]
    const float* w;
    int w_ldk, w_ldm;     // weight element (k, m) at w[k*w_ldk + m*w_ldm]
    float* y;             // [N][M][P]
    int N, K, M, P;
    int in_act;
    int strided;          // forward only: gather input at even (h, w)
    int T, H, W, Ho, Wo;  // strided: input H,W / output Ho,Wo.  addend_stride 2: output H,W / addend Ho,Wo
    long long Pin;        // input voxels per (n, k) row
    float* partial;       // [N][M][tiles][2]
    int tiles;
    const float* ex;      // EPI_ACTBWD: raw x [N][M][P]
    const float* ecoef;   // EPI_ACTBWD: [N][M][2]
    int e_act;
    const float* addend;  // EPI_PLAIN / EPI_ACTBWD (may be NULL)
    int addend_stride;
    int mblocks, mt_run;  // M blocks per voxel tile; 16-row tiles per block actually used
};

template <int MT, int IN, int EPI, bool VEC>
__global__ __launch_bounds__(256) void pw_kernel(const PwArgs A) {
    __shared__ float Wl[MT * 16 * PW_KPAD];
    __shared__ float red[4 * MT * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int n = blockIdx.y;
    const int mb = blockIdx.x % A.mblocks, tile = blockIdx.x / A.mblocks;
    const int mt_run = A.mt_run;
    const int m0 = mb * mt_run * 16;
    const int bm = min(mt_run * 16, A.M - m0);   // rows of this block that exist
    const int p0 = tile * PW_BN + wave * 64 + 4 * r;
    const int K = A.K, P = A.P;

    // per-lane input offsets of its 4 voxels (scalar path / strided gather)
    int off[4];
    bool pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = p0 + j;
        pv[j] = p < P;
        if (!VEC) {
            if (IN != IN_BNBWD && A.strided) {
                const int hw = A.Ho * A.Wo;
                const int t = p / hw, rem = p - t * hw;
                const int ho = rem / A.Wo, wo = rem - ho * A.Wo;
                off[j] = pv[j] ? (t * A.H + 2 * ho) * A.W + 2 * wo : 0;
            } else {
                off[j] = pv[j] ? p : 0;
            }
        } else {
            off[j] = p;
        }
    }

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = (K + 3) / 4;        // k-steps of 4 channels
    const int nchunks = (K + PW_KC - 1) / PW_KC;

    // register staging of one half chunk (4 k-steps = 16 channels)
    float4 rx[4], ra[4];
    float c0[4], c1[4], c2[4];

    auto load_half = [&](int hc) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = hc * 16 + ks * 4 + q;
            const bool kv = k < K;
            const int kc = kv ? k : 0;
            const float* px = A.x + ((size_t)n * K + kc) * (size_t)A.Pin;
            if (VEC) {
                rx[ks] = (kv && pv[0]) ? *reinterpret_cast<const float4*>(px + off[0]) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                rx[ks].x = (kv && pv[0]) ? px[off[0]] : 0.f;
                rx[ks].y = (kv && pv[1]) ? px[off[1]] : 0.f;
                rx[ks].z = (kv && pv[2]) ? px[off[2]] : 0.f;
                rx[ks].w = (kv && pv[3]) ? px[off[3]] : 0.f;
            }
            if (IN == IN_BNBWD) {
                const float* pa = A.a + ((size_t)n * K + kc) * (size_t)P;
                if (VEC) {
                    ra[ks] = (kv && pv[0]) ? *reinterpret_cast<const float4*>(pa + off[0]) : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    ra[ks].x = (kv && pv[0]) ? pa[off[0]] : 0.f;
                    ra[ks].y = (kv && pv[1]) ? pa[off[1]] : 0.f;
                    ra[ks].z = (kv && pv[2]) ? pa[off[2]] : 0.f;
                    ra[ks].w = (kv && pv[3]) ? pa[off[3]] : 0.f;
                }
                const float* pc = A.cin + ((size_t)n * K + kc) * 3;
                c0[ks] = kv ? pc[0] : 0.f;
                c1[ks] = kv ? pc[1] : 0.f;
                c2[ks] = kv ? pc[2] : 0.f;
            } else if (IN == IN_AFFACT) {
                const float* pc = A.cin + ((size_t)n * K + kc) * 2;
                c0[ks] = kv ? pc[0] : 0.f;
                c1[ks] = kv ? pc[1] : 0.f;
                c2[ks] = kv ? 1.f : 0.f;
            }
        }
    };

    float4 xb[4];
    auto combine_half = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 v = rx[ks];
            if (IN == IN_BNBWD) {
                const float4 av = ra[ks];
                v.x = pv[0] ? fmaf(c0[ks], v.x, fmaf(c1[ks], av.x, c2[ks])) : 0.f;
                v.y = pv[1] ? fmaf(c0[ks], v.y, fmaf(c1[ks], av.y, c2[ks])) : 0.f;
                v.z = pv[2] ? fmaf(c0[ks], v.z, fmaf(c1[ks], av.z, c2[ks])) : 0.f;
                v.w = pv[3] ? fmaf(c0[ks], v.w, fmaf(c1[ks], av.w, c2[ks])) : 0.f;
            } else if (IN == IN_AFFACT) {
                const bool kv = c2[ks] != 0.f;
                v.x = (kv && pv[0]) ? act_fwd(fmaf(c0[ks], v.x, c1[ks]), A.in_act) : 0.f;
                v.y = (kv && pv[1]) ? act_fwd(fmaf(c0[ks], v.y, c1[ks]), A.in_act) : 0.f;
                v.z = (kv && pv[2]) ? act_fwd(fmaf(c0[ks], v.z, c1[ks]), A.in_act) : 0.f;
                v.w = (kv && pv[3]) ? act_fwd(fmaf(c0[ks], v.w, c1[ks]), A.in_act) : 0.f;
            }
            xb[ks] = v;
        }
    };

    auto stage_w = [&](int c) {
        const int k0 = c * PW_KC;
        const int total = mt_run * 16 * PW_KC;
        if (A.w_ldk == 1) {   // forward layout: k contiguous in memory
            for (int idx = tid; idx < total; idx += 256) {
                const int m = idx >> 5, kk = idx & 31;
                float v = 0.f;
                if (m < bm && k0 + kk < K) v = A.w[(size_t)(m0 + m) * A.w_ldm + (k0 + kk)];
                Wl[m * PW_KPAD + kk] = v;
            }
        } else {              // transposed use (backward-data): m contiguous in memory
            const int rows = mt_run * 16;
            for (int idx = tid; idx < total; idx += 256) {
                const int kk = idx / rows, m = idx - kk * rows;
                float v = 0.f;
                if (m < bm && k0 + kk < K) v = A.w[(size_t)(k0 + kk) * A.w_ldk + (m0 + m)];
                Wl[m * PW_KPAD + kk] = v;
            }
        }
    };

    auto compute_half = [&](int half) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 b = xb[ks];
            const int kk = half * 16 + ks * 4 + q;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt < mt_run) {
                    const float av = Wl[(mt * 16 + r) * PW_KPAD + kk];
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, acc[mt][1], 0, 0, 0);
                    acc[mt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.z, acc[mt][2], 0, 0, 0);
                    acc[mt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.w, acc[mt][3], 0, 0, 0);
                }
            }
        }
    };

    const int nhalf = (nsteps + 3) / 4;
    load_half(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        stage_w(c);
        __syncthreads();
        const int h0 = 2 * c;
        combine_half();
        if (h0 + 1 < nhalf) load_half(h0 + 1);
        compute_half(0);
        if (h0 + 1 < nhalf) {
            combine_half();
            if (h0 + 2 < nhalf) load_half(h0 + 2);
            compute_half(1);
        }
    }

    // ------------------------------ epilogue ------------------------------
    int aoff[4];
    if (EPI != EPI_STATS && A.addend != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + j;
            if (A.addend_stride == 2) {
                const int hw = A.H * A.W;
                const int t = p / hw, rem = p - t * hw;
                const int h = rem / A.W, w = rem - h * A.W;
                aoff[j] = (pv[j] && !(h & 1) && !(w & 1)) ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : -1;
            } else {
                aoff[j] = pv[j] ? p : -1;
            }
        }
    }
    const long long addP = (A.addend_stride == 2) ? (long long)A.T * A.Ho * A.Wo : (long long)P;

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt < mt_run) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mt * 16 + 4 * q + e;
                const int m = m0 + ml;
                const bool mv = ml < bm;
                float4 v = make_float4(acc[mt][0][e], acc[mt][1][e], acc[mt][2][e], acc[mt][3][e]);
                float s1 = 0.f, s2 = 0.f;
                if (mv) {
                    float* py = A.y + ((size_t)n * A.M + m) * (size_t)P;
                    if (EPI != EPI_STATS && A.addend != nullptr) {
                        const float* pa = A.addend + ((size_t)n * A.M + m) * (size_t)addP;
                        if (aoff[0] >= 0) v.x += pa[aoff[0]];
                        if (aoff[1] >= 0) v.y += pa[aoff[1]];
                        if (aoff[2] >= 0) v.z += pa[aoff[2]];
                        if (aoff[3] >= 0) v.w += pa[aoff[3]];
                    }
                    if (EPI == EPI_ACTBWD) {
                        const float* px = A.ex + ((size_t)n * A.M + m) * (size_t)P;
                        const float sc = A.ecoef[((size_t)n * A.M + m) * 2], sh = A.ecoef[((size_t)n * A.M + m) * 2 + 1];
                        float4 xv;
                        if (VEC) {
                            xv = pv[0] ? *reinterpret_cast<const float4*>(px + p0) : make_float4(0.f, 0.f, 0.f, 0.f);
                        } else {
                            xv.x = pv[0] ? px[p0] : 0.f;
                            xv.y = pv[1] ? px[p0 + 1] : 0.f;
                            xv.z = pv[2] ? px[p0 + 2] : 0.f;
                            xv.w = pv[3] ? px[p0 + 3] : 0.f;
                        }
                        v.x = pv[0] ? v.x * act_bwd(fmaf(sc, xv.x, sh), A.e_act) : 0.f;
                        v.y = pv[1] ? v.y * act_bwd(fmaf(sc, xv.y, sh), A.e_act) : 0.f;
                        v.z = pv[2] ? v.z * act_bwd(fmaf(sc, xv.z, sh), A.e_act) : 0.f;
                        v.w = pv[3] ? v.w * act_bwd(fmaf(sc, xv.w, sh), A.e_act) : 0.f;
                        s1 = (v.x + v.y) + (v.z + v.w);
                        s2 = fmaf(v.x, xv.x, fmaf(v.y, xv.y, fmaf(v.z, xv.z, v.w * xv.w)));
                    } else if (EPI == EPI_STATS) {
                        s1 = (v.x + v.y) + (v.z + v.w);
                        s2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
                    }
                    if (VEC) {
                        if (pv[0]) *reinterpret_cast<float4*>(py + p0) = v;
                    } else {
                        if (pv[0]) py[p0] = v.x;
                        if (pv[1]) py[p0 + 1] = v.y;
                        if (pv[2]) py[p0 + 2] = v.z;
                        if (pv[3]) py[p0 + 3] = v.w;
                    }
                }
                if (EPI != EPI_PLAIN) {
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    if (r == 0) {
                        red[(wave * MT * 16 + ml) * 2] = s1;
                        red[(wave * MT * 16 + ml) * 2 + 1] = s2;
                    }
                }
            }
        }
    }
    if (EPI != EPI_PLAIN && A.partial != nullptr) {
        __syncthreads();
        for (int idx = tid; idx < bm * 2; idx += 256) {
            const int ml = idx >> 1, which = idx & 1;
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) s += red[(wv * MT * 16 + ml) * 2 + which];
            A.partial[(((size_t)n * A.M + (m0 + ml)) * A.tiles + tile) * 2 + which] = s;
        }
    }
}

template <int IN, int EPI>
int launch_pw(PwArgs& A, hipStream_t s) {
    const int mtiles = cdiv(A.M, 16);
    const int mblocks = cdiv(mtiles, 8);
    const int mt_run = cdiv(mtiles, mblocks);
    A.mblocks = mblocks;
    A.mt_run = mt_run;
    A.tiles = cdiv(A.P, PW_BN);
    const bool vec = (A.P % 4 == 0) && (A.Pin % 4 == 0) && !A.strided;
    dim3 grid(A.tiles * mblocks, A.N), block(256);
    if (mt_run <= 4) {
        if (vec) hipLaunchKernelGGL((pw_kernel<4, IN, EPI, true>), grid, block, 0, s, A);
        else hipLaunchKernelGGL((pw_kernel<4, IN, EPI, false>), grid, block, 0, s, A);
    } else {
        if (vec) hipLaunchKernelGGL((pw_kernel<8, IN, EPI, true>), grid, block, 0, s, A);
        else hipLaunchKernelGGL((pw_kernel<8, IN, EPI, false>), grid, block, 0, s, A);
    }
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

// ---------------------------------------------------------------------------------------
// Backward-weight: dW[co][ci] = sum_{n,p} dY[co][p] * in[ci][p].
// Each workgroup owns a 64x64 block of dW and a strided set of 128-voxel tiles; both
// operand tiles are staged in LDS as [channel][132] (built on the fly from g/a/x with the
// BN-backward combine and the forward prologue), each wave accumulates a 2x2 grid of 16x16
// MFMA tiles (K index = voxel), one ds_read_b128 feeding 4 MFMA steps.
// ---------------------------------------------------------------------------------------
constexpr int WG_PT = 128;
constexpr int WG_LD = 132;

struct WgArgs {
    const float* g; const float* a; const float* cb;      // [N][Co][P], [N][Co][P], [N][Co][3]
    const float* x; const float* pre; int pre_act;        // [N][Ci][Pin], [N][Ci][2] or NULL
    float* wpartial;                                       // [groups][Co][Ci]
    int N, Ci, Co, P; long long Pin;
    int strided, T, H, W, Ho, Wo;
    int groups, tiles_per_sample, cob, cib;
};

__global__ __launch_bounds__(256) void pw_wgrad_kernel(const WgArgs A) {
    __shared__ __attribute__((aligned(16))) float Ld[64 * WG_LD];
    __shared__ __attribute__((aligned(16))) float Lx[64 * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int blk = blockIdx.y;
    const int co0 = (blk / A.cib) * 64, ci0 = (blk % A.cib) * 64;
    const int wr = wave >> 1, wc = wave & 1;     // wave owns co tiles {2wr, 2wr+1} x ci tiles {2wc, 2wc+1}

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int total_tiles = A.N * A.tiles_per_sample;
    const bool vecd = (A.P % 4 == 0);
    const bool vecx = vecd && !A.strided && (A.Pin % 4 == 0);

    for (int tl = blockIdx.x; tl < total_tiles; tl += A.groups) {
        const int n = tl / A.tiles_per_sample, pt = (tl - n * A.tiles_per_sample) * WG_PT;
        __syncthreads();
        // stage dY tile: 64 rows x 128 voxels = 2048 float4 -> 8 per thread
        for (int idx = tid; idx < 64 * 32; idx += 256) {
            const int row = idx >> 5, c4 = (idx & 31) * 4;
            const int co = co0 + row, p = pt + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < A.Co && p < A.P) {
                const size_t base = ((size_t)n * A.Co + co) * (size_t)A.P + p;
                const float* cb = A.cb + ((size_t)n * A.Co + co) * 3;
                const float k0 = cb[0], k1 = cb[1], k2 = cb[2];
                if (vecd) {
                    const float4 gv = *reinterpret_cast<const float4*>(A.g + base);
                    const float4 av = *reinterpret_cast<const float4*>(A.a + base);
                    v.x = fmaf(k0, gv.x, fmaf(k1, av.x, k2));
                    v.y = fmaf(k0, gv.y, fmaf(k1, av.y, k2));
                    v.z = fmaf(k0, gv.z, fmaf(k1, av.z, k2));
                    v.w = fmaf(k0, gv.w, fmaf(k1, av.w, k2));
                } else {
                    float t4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        t4[j] = (p + j < A.P) ? fmaf(k0, A.g[base + j], fmaf(k1, A.a[base + j], k2)) : 0.f;
                    v = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
            }
            *reinterpret_cast<float4*>(&Ld[row * WG_LD + c4]) = v;
        }
        for (int idx = tid; idx < 64 * 32; idx += 256) {
            const int row = idx >> 5, c4 = (idx & 31) * 4;
            const int ci = ci0 + row, p = pt + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < A.Ci && p < A.P) {
                const float* px = A.x + ((size_t)n * A.Ci + ci) * (size_t)A.Pin;
                float t4[4];
                if (vecx) {
                    const float4 xv = *reinterpret_cast<const float4*>(px + p);
                    t4[0] = xv.x; t4[1] = xv.y; t4[2] = xv.z; t4[3] = xv.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pp = p + j;
                        float xv = 0.f;
                        if (pp < A.P) {
                            if (A.strided) {
                                const int hw = A.Ho * A.Wo;
                                const int t = pp / hw, rem = pp - t * hw;
                                const int ho = rem / A.Wo, wo = rem - ho * A.Wo;
                                xv = px[(size_t)(t * A.H + 2 * ho) * A.W + 2 * wo];
                            } else {
                                xv = px[pp];
                            }
                        }
                        t4[j] = xv;
                    }
                }
                if (A.pre != nullptr) {
                    const float sc = A.pre[((size_t)n * A.Ci + ci) * 2], sh = A.pre[((size_t)n * A.Ci + ci) * 2 + 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) t4[j] = (p + j < A.P) ? act_fwd(fmaf(sc, t4[j], sh), A.pre_act) : 0.f;
                }
                v = make_float4(t4[0], t4[1], t4[2], t4[3]);
            }
            *reinterpret_cast<float4*>(&Lx[row * WG_LD + c4]) = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < WG_PT / 16; ++kk) {
            float4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = *reinterpret_cast<const float4*>(&Ld[((2 * wr + i) * 16 + r) * WG_LD + kk * 16 + 4 * q]);
                bv[i] = *reinterpret_cast<const float4*>(&Lx[((2 * wc + i) * 16 + r) * WG_LD + kk * 16 + 4 * q]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    // D[i = co][j = ci]: lane (q, r), reg e -> co = tile*16 + 4q + e, ci = tile*16 + r
    float* out = A.wpartial + (size_t)blockIdx.x * A.Co * A.Ci;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + (2 * wr + i) * 16 + 4 * q + e;
                const int ci = ci0 + (2 * wc + j) * 16 + r;
                if (co < A.Co && ci < A.Ci) out[(size_t)co * A.Ci + ci] = acc[i][j][e];
            }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial,
                                                              float* __restrict__ out, int groups, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int g = 0; g < groups; ++g) s += (double)partial[(size_t)g * n + i];
    out[i] = (float)s;
}

}  // namespace

extern "C" int x3d_pw_tiles(int P) { return cdiv(P, PW_BN); }

extern "C" int x3d_pw_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout, int T, int H,
                          int W, int strideHW, const float* pre, int pre_act, float* partial, void* stream) {
    X3D_CHECK_ARG(x && w && y);
    X3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    X3D_CHECK_ARG(N <= 65535);
    PwArgs A = {};
    const int Ho = strideHW == 2 ? (H - 1) / 2 + 1 : H, Wo = strideHW == 2 ? (W - 1) / 2 + 1 : W;
    A.x = x; A.a = nullptr; A.cin = pre; A.w = w; A.w_ldk = 1; A.w_ldm = Cin; A.y = y;
    A.N = N; A.K = Cin; A.M = Cout; A.P = T * Ho * Wo; A.Pin = (long long)T * H * W;
    A.in_act = pre_act; A.strided = strideHW == 2; A.T = T; A.H = H; A.W = W; A.Ho = Ho; A.Wo = Wo;
    A.partial = partial; A.addend = nullptr; A.addend_stride = 1;
    hipStream_t s = (hipStream_t)stream;
    if (pre) return launch_pw<IN_AFFACT, EPI_STATS>(A, s);
    return launch_pw<IN_RAW, EPI_STATS>(A, s);
}

extern "C" int x3d_pw_bwd_data(const float* g, const float* a, const float* cb, const float* w, float* out,
                               int N, int Cin, int Cout, int T, int H, int W, const float* x,
                               const float* pre, int pre_act, const float* addend, int addend_stride,
                               float* partial, void* stream) {
    X3D_CHECK_ARG(g && a && cb && w && out);
    X3D_CHECK_ARG(N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(addend_stride == 1 || addend_stride == 2);
    X3D_CHECK_ARG((pre == nullptr) || (x != nullptr));
    PwArgs A = {};
    A.x = g; A.a = a; A.cin = cb; A.w = w; A.w_ldk = Cin; A.w_ldm = 1; A.y = out;
    A.N = N; A.K = Cout; A.M = Cin; A.P = T * H * W; A.Pin = A.P;
    A.strided = 0; A.T = T; A.H = H; A.W = W;
    A.Ho = (H - 1) / 2 + 1; A.Wo = (W - 1) / 2 + 1;
    A.partial = partial; A.ex = x; A.ecoef = pre; A.e_act = pre_act;
    A.addend = addend; A.addend_stride = addend_stride;
    hipStream_t s = (hipStream_t)stream;
    if (pre) return launch_pw<IN_BNBWD, EPI_ACTBWD>(A, s);
    return launch_pw<IN_BNBWD, EPI_PLAIN>(A, s);
}

extern "C" int x3d_pw_wgrad_groups(int N, int P) {
    const int tiles = N * cdiv(P, WG_PT);
    return tiles < 512 ? tiles : 512;
}

extern "C" int x3d_pw_bwd_weight(const float* g, const float* a, const float* cb, const float* x,
                                 const float* pre, int pre_act, float* wpartial, int N, int Cin, int Cout,
                                 int T, int H, int W, int strideHW, void* stream) {
    X3D_CHECK_ARG(g && a && cb && x && wpartial);
    X3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    WgArgs A = {};
    const int Ho = strideHW == 2 ? (H - 1) / 2 + 1 : H, Wo = strideHW == 2 ? (W - 1) / 2 + 1 : W;
    A.g = g; A.a = a; A.cb = cb; A.x = x; A.pre = pre; A.pre_act = pre_act; A.wpartial = wpartial;
    A.N = N; A.Ci = Cin; A.Co = Cout; A.P = T * Ho * Wo; A.Pin = (long long)T * H * W;
    A.strided = strideHW == 2; A.T = T; A.H = H; A.W = W; A.Ho = Ho; A.Wo = Wo;
    A.tiles_per_sample = cdiv(A.P, WG_PT);
    A.groups = x3d_pw_wgrad_groups(N, A.P);
    A.cob = cdiv(Cout, 64); A.cib = cdiv(Cin, 64);
    X3D_CHECK_ARG(A.cob * A.cib <= 65535);
    dim3 grid(A.groups, A.cob * A.cib), block(256);
    hipLaunchKernelGGL(pw_wgrad_kernel, grid, block, 0, (hipStream_t)stream, A);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_reduce_partials(const float* partial, float* out, int groups, int n, void* stream) {
    X3D_CHECK_ARG(partial && out && groups > 0 && n > 0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       partial, out, groups, n);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
