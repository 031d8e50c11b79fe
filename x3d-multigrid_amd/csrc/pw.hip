// Pointwise (1x1x1) convolution kernels: MFMA GEMMs with fused prologues/epilogues.
//
// Reference call sites replaced: conv1x1x1 (x3d.py:98-103) used as Bottleneck.conv1/conv3
// (:112,116,146,162), downsample[0] (:272), conv5 (:231,327) and their autograd backward.
//
// Kernels in this file (DESIGN.md 4.1 has the table):
//   pw3_kernel        forward / data-gradient, stages 1-2: barrier-free streaming, activations straight
//                     from HBM as one float4 (or dword) per lane, pre-packed weight fragments, fp32 MFMA
//   pw4_kernel        forward, stages 3-4: persistent software-pipelined LDS-tiled GEMM, fp32 MFMA
//   pw5_kernel        data-gradient, stages 3-4: pw4's pipeline on split-bf16 MFMA (hi+lo, 3 products)
//   pw2_kernel, pw_kernel   fallbacks (P % 4 != 0, strided gather, no packed weights)
//   pw_wgrad3_kernel  weight gradient, every dense shape: split-bf16 MFMA, LDS-tiled, XCD-aware groups
//   pw_wgrad2_kernel, pw_wgrad_kernel   exact-fp32 / strided weight gradient
//   pw_pack*_kernel   weights -> MFMA fragment order (fp32 image; transposed image also as bf16 hi/lo planes)
// Common mapping: output channels on the MFMA row index, voxels on the column index, so every lane
// owns consecutive voxels of a channel row and results are stored as float4 / float2 row segments.
// Forward GEMMs are exact fp32 (v_mfma_f32_16x16x4_f32, the same rounding as a VALU fmaf chain); the
// backward GEMMs use v_mfma_f32_16x16x32_bf16 on split operands (DESIGN.md 4.2 for why only there).
// BN statistics / BN-backward reductions ride in the epilogues: 16-lane DPP row sums, then
// one partial per (sample, channel, voxel tile) -- summed later in fixed order (fp64).
#include <cstdlib>
#include "common.h"

// pw6.hip: forward kernel of the large-channel layers (whole-K items, 3-term split-bf16 MFMA)
bool x3d_pw6_ok(int K, int M, int P);
int x3d_pw6_tiles(int P);
int x3d_pw6_launch(const void* x, const float* cin, const float* wp, void* y, float* partial, int N, int K, int M,
                   int P, int in_act, int x_bf, int y_bf, hipStream_t s);
bool x3d_pwfs_ok(int K, int M, int P);
int x3d_pwfs_tiles(int N, int P);
int x3d_pwfs_launch(const void* x, const float* cin, const float* wp, void* y, float* partial, int N, int K, int M, int P,
                    int in_act, int x_bf, int y_bf, hipStream_t s);
bool x3d_pw7_ok(int K, int M, int P, int mx);
int x3d_pw7_launch(const void* g, const void* a, const float* cb, const float* wpt, void* out, float* partial, int mode,
                   const void* ex, const float* emask, const float* ecoef, int e_act, const float* addend,
                   int addend_stride, int N, int K, int M, int T, int H, int W, int ga_bf, int y_bf, int ex_bf, hipStream_t s);

namespace {

enum { IN_RAW = 0, IN_AFFACT = 1, IN_BNBWD = 2 };
// EPI_RESBWD: the residual-add + ReLU backward of the block that produced this conv's input, folded into the data
// gradient's epilogue:  y = (acc + addend) where emask > 0 else 0;  statistics sum(y), sum(y * ex)  (ex = that block's raw
// conv3 output, emask = its output) -- what bn_add_relu_bwd computes in a launch of its own (x3d.py:165-169 backward).
enum { EPI_STATS = 0, EPI_PLAIN = 1, EPI_ACTBWD = 2, EPI_RESBWD = 3 };
#define EPI_HAS_X(E) ((E) == EPI_ACTBWD || (E) == EPI_RESBWD)
constexpr int PW_MAXK = 640;   // per-sample input-coefficient table in LDS (XL: 630 channels)


struct PwArgs {
    const float* x;       // IN_RAW / IN_AFFACT: input [N][K][Pin];  IN_BNBWD: upstream grad g [N][K][P]
    const float* a;       // IN_BNBWD: raw forward output [N][K][P]
    const float* cin;     // IN_AFFACT: [N][K][2];  IN_BNBWD: [N][K][3]
    const float* w;
    const float* wp;      // weights pre-packed by x3d_pw_pack (tiled variant); may be NULL
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
    const float* emask;   // EPI_RESBWD: [N][M][P], gradient passes where emask > 0 (ex is then the statistics' multiplier)
    int e_act;
    const float* addend;  // EPI_PLAIN / EPI_ACTBWD (may be NULL)
    int addend_stride;
    int mblocks, mt_run;  // M blocks per voxel tile; 16-row tiles per block actually used
    // mixed-storage mode (kernels built with MX only; the pointers above are then bf16 arrays behind a float* type):
    int x_bf;             // x (and a) are bf16
    int y_bf;             // y is bf16
    int ex_bf;            // ex is bf16
};

// NT consecutive voxels per lane: NT = 4 -> float4 loads/stores (needs P % 4 == 0, dense input),
// NT = 1 -> dword accesses (any P, strided gather, and 4x more waves for small-P layers).
template <int NT> struct VecT;
template <> struct VecT<4> { typedef float4 type; };
template <> struct VecT<1> { typedef float type; };

template <int NT>
__device__ __forceinline__ void vload(const float* p, float (&v)[NT]) {
    if (NT == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1 % NT] = t.y; v[2 % NT] = t.z; v[3 % NT] = t.w;
    } else {
        v[0] = p[0];
    }
}
template <int NT>
__device__ __forceinline__ void vstore(float* p, const float (&v)[NT]) {
    if (NT == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % NT], v[2 % NT], v[3 % NT]);
    else p[0] = v[0];
}

// mixed-storage forms: wave-uniform byte base + element offset, or pointer + element index
// (raw: the payload is widened later by vwiden -- see ldx4_raw in common.h)
template <int NT>
__device__ __forceinline__ void vloado(const char* base, unsigned eoff, int bf, float (&v)[NT]) {
    if (NT == 4) {
        const float4 t = ldo4_raw(base, eoff, bf);
        v[0] = t.x; v[1 % NT] = t.y; v[2 % NT] = t.z; v[3 % NT] = t.w;
    } else {
        v[0] = ldo1_raw(base, eoff, bf);
    }
}
template <int NT>
__device__ __forceinline__ void vloadx(const void* p, size_t i, int bf, float (&v)[NT]) {
    if (NT == 4) {
        const float4 t = ldx4_raw(p, i, bf);
        v[0] = t.x; v[1 % NT] = t.y; v[2 % NT] = t.z; v[3 % NT] = t.w;
    } else {
        v[0] = ldo1_raw(reinterpret_cast<const char*>(p) + i * (bf ? 2u : 4u), 0u, bf);
    }
}
template <int NT>
__device__ __forceinline__ void vwiden(const float (&r)[NT], int bf, float (&v)[NT]) {
    if (NT == 4) {
        const float4 t = widen4(make_float4(r[0], r[1 % NT], r[2 % NT], r[3 % NT]), bf);
        v[0] = t.x; v[1 % NT] = t.y; v[2 % NT] = t.z; v[3 % NT] = t.w;
    } else {
        v[0] = widen1(r[0], bf);
    }
}
template <int NT>
__device__ __forceinline__ void vstorex(void* p, size_t i, int bf, const float (&v)[NT]) {
    if (NT == 4) stx4(p, i, bf, v[0], v[1 % NT], v[2 % NT], v[3 % NT]);
    else stx1(p, i, bf, v[0]);
}

template <int MT, int NT, int IN, int EPI>
__global__ __launch_bounds__(256, 2) void pw_kernel(const PwArgs A) {
    constexpr int PW_KC = 32;     // channels per LDS weight chunk
    constexpr int PW_KPAD = 34;   // LDS row stride of the weight chunk (conflict-free fragment reads)
    __shared__ float Wl[MT * 16 * PW_KPAD];
    __shared__ float red[4 * MT * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int n = blockIdx.y;
    // block order: the M-blocks of one voxel tile get ids 8 apart, i.e. (round-robin dispatch)
    // the same XCD, so the activation tile they all read is fetched into that XCD's L2 once
    const int tlo = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int mb = rest % A.mblocks, tile = (rest / A.mblocks) * 8 + tlo;
    if (tile >= A.tiles) return;
    const int mt_run = A.mt_run;
    const int m0 = mb * mt_run * 16;
    const int bm = min(mt_run * 16, A.M - m0);   // rows of this block that exist
    const int p0 = tile * (64 * NT) + wave * (16 * NT) + NT * r;
    const int K = A.K, P = A.P;
    const bool pv = p0 < P;                      // NT == 4 implies P % 4 == 0: all-or-nothing

    int off = p0;                                // input offset of this lane's first voxel
    if (NT == 1 && IN != IN_BNBWD && A.strided && pv) {
        const int hw = A.Ho * A.Wo;
        const int t = p0 / hw, rem = p0 - t * hw;
        const int ho = rem / A.Wo, wo = rem - ho * A.Wo;
        off = (t * A.H + 2 * ho) * A.W + 2 * wo;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = (K + 3) / 4;        // k-steps of 4 channels
    const int nchunks = (K + PW_KC - 1) / PW_KC;

    // register staging of one half chunk (4 k-steps = 16 channels)
    float rx[4][NT], ra[4][NT];
    float c0[4], c1[4], c2[4];

    auto load_half = [&](int hc) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = hc * 16 + ks * 4 + q;
            const bool kv = k < K;
            const int kc = kv ? k : 0;
#pragma unroll
            for (int j = 0; j < NT; ++j) { rx[ks][j] = 0.f; ra[ks][j] = 0.f; }
            if (kv && pv) vload<NT>(A.x + ((size_t)n * K + kc) * (size_t)A.Pin + off, rx[ks]);
            if (IN == IN_BNBWD) {
                if (kv && pv) vload<NT>(A.a + ((size_t)n * K + kc) * (size_t)P + off, ra[ks]);
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

    float xb[4][NT];
    auto combine_half = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v = rx[ks][j];
                if (IN == IN_BNBWD) {
                    v = pv ? fmaf(c0[ks], v, fmaf(c1[ks], ra[ks][j], c2[ks])) : 0.f;
                } else if (IN == IN_AFFACT) {
                    v = (c2[ks] != 0.f && pv) ? act_fwd(fmaf(c0[ks], v, c1[ks]), A.in_act) : 0.f;
                }
                xb[ks][j] = v;
            }
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
            const int kk = half * 16 + ks * 4 + q;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt < mt_run) {
                    const float av = Wl[(mt * 16 + r) * PW_KPAD + kk];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[ks][j], acc[mt][j], 0, 0, 0);
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
    // Per 16-row tile: phase 1 issues the four rows' global reads (activation-derivative input,
    // its coefficients, the residual addend) from clamped addresses, phase 2 combines and stores
    // -- no load sits under a per-lane branch (see pw2_kernel).
    const bool has_add = EPI != EPI_STATS && A.addend != nullptr;
    const bool add_s2 = has_add && A.addend_stride == 2;
    int aoff[NT];
    bool av[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int p = min(p0 + j, P - 1);
        av[j] = has_add && pv;
        aoff[j] = p;
        if (add_s2) {
            const int hw = A.H * A.W;
            const int t = p / hw, rem = p - t * hw;
            const int h = rem / A.W, w = rem - h * A.W;
            const bool even = !(h & 1) && !(w & 1);
            av[j] = av[j] && even;
            aoff[j] = even ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : 0;
        }
    }
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;
    const int pc = pv ? p0 : 0;                   // clamped voxel (NT-aligned)

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt < mt_run) {
            float xv[4][NT], mk[4][NT], adv[4][NT], esc[4], esh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mt * 16 + 4 * q + e;
                const size_t mrow = (size_t)n * A.M + m0 + (ml < bm ? ml : 0);
                if (EPI == EPI_ACTBWD) {
                    const float2 c2 = *reinterpret_cast<const float2*>(A.ecoef + mrow * 2);
                    esc[e] = c2.x; esh[e] = c2.y;
                }
                if (EPI_HAS_X(EPI)) vload<NT>(A.ex + mrow * (size_t)P + pc, xv[e]);
                if (EPI == EPI_RESBWD) vload<NT>(A.emask + mrow * (size_t)P + pc, mk[e]);
                if (has_add) {
                    const float* pa = A.addend + mrow * (size_t)addP;
                    if (NT == 4 && !add_s2) {
                        vload<NT>(pa + pc, adv[e]);
                    } else {
#pragma unroll
                        for (int j = 0; j < NT; ++j) adv[e][j] = pa[aoff[j]];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mt * 16 + 4 * q + e;
                const int m = m0 + ml;
                const bool mv = ml < bm;
                float v[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) v[j] = acc[mt][j][e];
                float s1 = 0.f, s2 = 0.f;
                if (has_add) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) v[j] += av[j] ? adv[e][j] : 0.f;
                }
                if (EPI_HAS_X(EPI)) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float xj = pv ? xv[e][j] : 0.f;
                        if (EPI == EPI_RESBWD) v[j] = (pv && mk[e][j] > 0.f) ? v[j] : 0.f;
                        else v[j] = pv ? v[j] * act_bwd(fmaf(esc[e], xj, esh[e]), A.e_act) : 0.f;
                        s1 += v[j];
                        s2 = fmaf(v[j], xj, s2);
                    }
                } else if (EPI == EPI_STATS) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) { v[j] = pv ? v[j] : 0.f; s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
                }
                if (mv && pv) vstore<NT>(A.y + ((size_t)n * A.M + m) * (size_t)P + p0, v);
                if (EPI != EPI_PLAIN) {
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    if (r == 0) {
                        red[(wave * MT * 16 + ml) * 2] = mv ? s1 : 0.f;
                        red[(wave * MT * 16 + ml) * 2 + 1] = mv ? s2 : 0.f;
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

// ---------------------------------------------------------------------------------------
// Streaming variant with pre-packed weights (the default for stages 1-2): no LDS staging and no
// barrier in the K loop.  Per group of 16 channels a wave issues MT coalesced 1-KiB loads of
// packed weight fragments (L2-resident) and 4 activation loads (NT voxels x 4 channel rows
// each, straight from HBM), then MT*4*NT MFMAs; the next group's loads are in flight meanwhile.
// Channels >= K meet zero-padded weights and voxels >= P are never stored, so no load needs a
// predicate -- addresses are only clamped into the tensor.
// ---------------------------------------------------------------------------------------
template <int MT, int NT, int IN, int EPI, bool MX>
__global__ __launch_bounds__(256, 2) void pw3_kernel(const PwArgs A) {
    const int x_bf = MX ? A.x_bf : 0, y_bf = MX ? A.y_bf : 0, ex_bf = MX ? A.ex_bf : 0;      // fp32 build: folded away
    __shared__ float red[4 * MT * 16 * 2];
    __shared__ float Cl[(IN == IN_RAW) ? 4 : 3 * PW_MAXK];
    constexpr int NC = (IN == IN_BNBWD) ? 3 : 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int n = blockIdx.y;
    const int tlo = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int mb = rest % A.mblocks, tile = (rest / A.mblocks) * 8 + tlo;
    if (tile >= A.tiles) return;
    const int mt_run = A.mt_run;
    const int m0 = mb * mt_run * 16;
    const int bm = min(mt_run * 16, A.M - m0);
    const int p0 = tile * (64 * NT) + wave * (16 * NT) + NT * r;
    const int K = A.K, P = A.P;
    const bool pv = p0 < P;
    const int kgroups = (K + 15) / 16;

    int off = pv ? p0 : 0;                       // clamped: always a valid voxel
    if (NT == 1 && IN != IN_BNBWD && A.strided && pv) {
        const int hw = A.Ho * A.Wo;
        const int t = p0 / hw, rem = p0 - t * hw;
        const int ho = rem / A.Wo, wo = rem - ho * A.Wo;
        off = (t * A.H + 2 * ho) * A.W + 2 * wo;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-sample bases are wave uniform; a sample's rows are < 2^32 elements away
    const char* xs = mx_base(A.x, (size_t)n * K * (size_t)A.Pin, x_bf);
    const char* as = (IN == IN_BNBWD) ? mx_base(A.a, (size_t)n * K * (size_t)P, x_bf) : nullptr;
    const unsigned Pin32 = (unsigned)A.Pin;
    const float* wpl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        wpl[mt] = A.wp + ((size_t)min(mb * mt_run + mt, (A.M + 15) / 16 - 1) * kgroups) * 256 + lane * 4;   // clamped to a packed tile

    float4 wa[MT], wn[MT];
    float xb[4][NT], xn[4][NT], ab[(IN == IN_BNBWD) ? 4 : 1][NT], an_[(IN == IN_BNBWD) ? 4 : 1][NT];

    auto issue = [&](int s, float4 (&wd)[MT], float (&xd)[4][NT], float (&ad)[(IN == IN_BNBWD) ? 4 : 1][NT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wd[mt] = *reinterpret_cast<const float4*>(wpl[mt] + (size_t)s * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = min(16 * s + 4 * q + e, K - 1);
            vloado<NT>(xs, (unsigned)k * Pin32 + (unsigned)off, x_bf, xd[e]);
            if (IN == IN_BNBWD) vloado<NT>(as, (unsigned)k * (unsigned)P + (unsigned)off, x_bf, ad[e]);
        }
    };

    auto compute = [&](int s, const float4 (&wd)[MT], const float (&xd)[4][NT],
                       const float (&ad)[(IN == IN_BNBWD) ? 4 : 1][NT]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = min(16 * s + 4 * q + e, K - 1);
            float b[NT], xw[NT], aw[NT];
            vwiden<NT>(xd[e], x_bf, xw);
            if (IN == IN_BNBWD) {
                vwiden<NT>(ad[e], x_bf, aw);
                const float c0 = Cl[k * 3], c1 = Cl[k * 3 + 1], c2 = Cl[k * 3 + 2];
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = fmaf(c0, xw[j], fmaf(c1, aw[j], c2));
            } else if (IN == IN_AFFACT) {
                const float sc = Cl[k * 2], sh = Cl[k * 2 + 1];
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = act_fwd(fmaf(sc, xw[j], sh), A.in_act);
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = xw[j];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float av = e == 0 ? wd[mt].x : (e == 1 ? wd[mt].y : (e == 2 ? wd[mt].z : wd[mt].w));
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j], acc[mt][j], 0, 0, 0);
            }
        }
    };

    issue(0, wa, xb, ab);          // in flight while the coefficient table is filled
    if (IN != IN_RAW) {
        for (int i = tid; i < K * NC; i += 256) Cl[i] = A.cin[(size_t)n * K * NC + i];
        __syncthreads();
    }
    for (int s = 0; s < kgroups; s += 2) {
        if (s + 1 < kgroups) issue(s + 1, wn, xn, an_);
        compute(s, wa, xb, ab);
        if (s + 1 < kgroups) {
            if (s + 2 < kgroups) issue(s + 2, wa, xb, ab);
            compute(s + 1, wn, xn, an_);
        }
    }

    // ------------------------------ epilogue ------------------------------
    // Per 16-row tile: phase 1 issues the four rows' global reads (activation-derivative input,
    // its coefficients, the residual addend) from clamped addresses, phase 2 combines and stores
    // -- no load sits under a per-lane branch (see pw2_kernel).
    const bool has_add = EPI != EPI_STATS && A.addend != nullptr;
    const bool add_s2 = has_add && A.addend_stride == 2;
    int aoff[NT];
    bool av[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int p = min(p0 + j, P - 1);
        av[j] = has_add && pv;
        aoff[j] = p;
        if (add_s2) {
            const int hw = A.H * A.W;
            const int t = p / hw, rem = p - t * hw;
            const int h = rem / A.W, w = rem - h * A.W;
            const bool even = !(h & 1) && !(w & 1);
            av[j] = av[j] && even;
            aoff[j] = even ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : 0;
        }
    }
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;
    const int pc = pv ? p0 : 0;                   // clamped voxel (NT-aligned)

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt < mt_run) {
            float xv[4][NT], mk[4][NT], adv[4][NT], esc[4], esh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mt * 16 + 4 * q + e;
                const size_t mrow = (size_t)n * A.M + m0 + (ml < bm ? ml : 0);
                if (EPI == EPI_ACTBWD) {
                    const float2 c2 = *reinterpret_cast<const float2*>(A.ecoef + mrow * 2);
                    esc[e] = c2.x; esh[e] = c2.y;
                }
                if (EPI_HAS_X(EPI)) vloadx<NT>(A.ex, mrow * (size_t)P + pc, EPI == EPI_ACTBWD ? ex_bf : 0, xv[e]);
                if (EPI == EPI_RESBWD) vload<NT>(A.emask + mrow * (size_t)P + pc, mk[e]);
                if (has_add) {
                    const float* pa = A.addend + mrow * (size_t)addP;
                    if (NT == 4 && !add_s2) {
                        vload<NT>(pa + pc, adv[e]);
                    } else {
#pragma unroll
                        for (int j = 0; j < NT; ++j) adv[e][j] = pa[aoff[j]];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mt * 16 + 4 * q + e;
                const int m = m0 + ml;
                const bool mv = ml < bm;
                if (EPI == EPI_ACTBWD) vwiden<NT>(xv[e], ex_bf, xv[e]);
                float v[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) v[j] = acc[mt][j][e];
                float s1 = 0.f, s2 = 0.f;
                if (has_add) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) v[j] += av[j] ? adv[e][j] : 0.f;
                }
                if (EPI_HAS_X(EPI)) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float xj = pv ? xv[e][j] : 0.f;
                        if (EPI == EPI_RESBWD) v[j] = (pv && mk[e][j] > 0.f) ? v[j] : 0.f;
                        else v[j] = pv ? v[j] * act_bwd(fmaf(esc[e], xj, esh[e]), A.e_act) : 0.f;
                        v[j] = stored(v[j], y_bf);
                        s1 += v[j];
                        s2 = fmaf(v[j], xj, s2);
                    }
                } else if (EPI == EPI_STATS) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) { v[j] = stored(pv ? v[j] : 0.f, y_bf); s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
                }
                if (mv && pv) vstorex<NT>(A.y, ((size_t)n * A.M + m) * (size_t)P + p0, y_bf, v);
                if (EPI != EPI_PLAIN) {
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    if (r == 0) {
                        red[(wave * MT * 16 + ml) * 2] = mv ? s1 : 0.f;
                        red[(wave * MT * 16 + ml) * 2 + 1] = mv ? s2 : 0.f;
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

// ---------------------------------------------------------------------------------------
// Tiled variant for the large-C / small-P layers (stages 3-4, conv5): K >= 64, M >= 96.
// Workgroup tile = up to 128 output channels x 64 voxels, K in chunks of 64 channels.
//   * activations: staged ONCE per workgroup into a double-buffered LDS tile [64 ch][64 voxels]
//     (global -> registers before the MFMAs of the current chunk, registers -> LDS after them,
//     one barrier per chunk) with the prologue (BN apply / BN-backward combine / Swish) applied
//     while staging; all four waves read the same fragments (ds_read_b128, 4 voxel tiles each);
//   * weights: pre-packed once per step (x3d_pw_pack) into MFMA fragment order
//     Wp[m-tile][k-group of 16][lane][4], zero padded, so a wave's A fragment for four k-steps is
//     ONE fully coalesced 1-KiB global_load_dwordx4 from L2 with no bounds logic and no LDS.
//   k index permutation: step e of lane group q uses channel 16*s + 4*q + e on both operands.
// ---------------------------------------------------------------------------------------
constexpr int P2_BN = 64, P2_KC = 64;
constexpr int P2_NB = P2_KC * P2_BN / 4 / 256;      // activation float4 slots per thread

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// Packed weight image.
//   fp32 part (always):   wp[((mt * kgroups + s) * 64 + lane) * 4 + e] = A[16 mt + (lane & 15)][16 s + 4 (lane >> 4) + e]
//   split-bf16 part, directly behind the fp32 part, for the 16x16x32 bf16 MFMA: three planes
//   (hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid): all 24 significant bits; a two-term kernel reads hi, mid) of
//                         wq[((mt * kg32 + s) * 64 + lane) * 8 + j] = A[16 mt + (lane & 15)][32 s + 8 (lane >> 4) + j]
// One work item per fp32 element, then one per bf16 (hi, lo) pair.
__device__ __forceinline__ void pack_item(int i, const float* __restrict__ w, float* __restrict__ wp, int M, int K, int ldm,
                                          int ldk, int mtiles, int kgroups, int with_bf16) {
    const int nf = mtiles * kgroups * 256;
    if (i < nf) {
        const int e = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
        const int s = blk % kgroups, mt = blk / kgroups;
        const int row = 16 * mt + (lane & 15), k = 16 * s + 4 * (lane >> 4) + e;
        wp[i] = (row < M && k < K) ? w[(size_t)row * ldm + (size_t)k * ldk] : 0.f;
        return;
    }
    if (!with_bf16) return;
    const int kg32 = (K + 31) / 32;
    const int nq = mtiles * kg32 * 512;
    const int t = i - nf;
    if (t >= nq) return;
    const int j = t & 7, lane = (t >> 3) & 63, blk = t >> 9;
    const int s = blk % kg32, mt = blk / kg32;
    const int row = 16 * mt + (lane & 15), k = 32 * s + 8 * (lane >> 4) + j;
    const float v = (row < M && k < K) ? w[(size_t)row * ldm + (size_t)k * ldk] : 0.f;
    const __bf16 h = (__bf16)v;
    __bf16* wq = reinterpret_cast<__bf16*>(wp + nf);
    wq[t] = h;
    // ALWAYS three planes hi + mid + lo = all 24 significant bits (every consumer since ABI 6 -- pw6 / pw7 / pwf / the
    // weight-gradient kernels -- reads the third plane; a job that still asked for two planes used to leave it
    // uninitialised: ADVICE r03).  A two-term kernel reads hi and mid.
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    wq[nq + t] = m;
    wq[2 * nq + t] = (__bf16)(r1 - (float)m);
}

// planes: 3 (hi, mid, lo) in both orientations since ABI 6
static size_t pack_items(int K, int M, int planes) {
    const size_t mt = cdiv(M, 16);
    return mt * cdiv(K, 16) * 256 + (planes ? mt * cdiv(K, 32) * 512 : 0);
}
static size_t pack_floats(int K, int M, int planes) {
    const size_t mt = cdiv(M, 16);
    return mt * cdiv(K, 16) * 256 + mt * cdiv(K, 32) * 256 * (size_t)planes;        // a bf16 plane = 256 floats per (tile, k step)
}

__global__ __launch_bounds__(256) void pw_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int M, int K,
                                                      int ldm, int ldk, int mtiles, int kgroups, int with_bf16) {
    pack_item(blockIdx.x * 256 + threadIdx.x, w, wp, M, K, ldm, ldk, mtiles, kgroups, with_bf16);
}

// Batched form: one launch packs every pointwise weight of the network in both orientations.
// jobs[] and the workgroup -> job table are built once on the host (x3dhip/engine.py).
struct PackJob {
    const float* w; float* wp;
    int M, K, ldm, ldk, mtiles, kgroups, wg0, with_bf16;
};

__global__ __launch_bounds__(256) void pw_pack_batch_kernel(const PackJob* __restrict__ jobs,
                                                            const int* __restrict__ wg_job) {
    const PackJob J = jobs[wg_job[blockIdx.x]];
    pack_item((blockIdx.x - J.wg0) * 256 + threadIdx.x, J.w, J.wp, J.M, J.K, J.ldm, J.ldk, J.mtiles, J.kgroups, J.with_bf16);
}

#ifdef X3D_TRACE
__device__ unsigned long long g_trace[16384 * 8];
extern "C" int x3d_debug_trace(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), bytes);
}
#define TR(i) do { if (tid == 0) trv[i] = wall_clock64(); } while (0)
#else
#define TR(i) do { } while (0)
#endif

template <int IN, int EPI, bool VEC, bool TWO>
__global__ __launch_bounds__(256, 2) void pw2_kernel(const PwArgs A) {
#ifdef X3D_TRACE
    unsigned long long trv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    __shared__ __attribute__((aligned(16))) float Bl[2][P2_KC * P2_BN];
    __shared__ float Cl[(IN == IN_RAW) ? 4 : 3 * PW_MAXK];
    constexpr int NC = (IN == IN_BNBWD) ? 3 : 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int n = blockIdx.y;
    const int tlo = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int mb = rest % A.mblocks, tile = (rest / A.mblocks) * 8 + tlo;
    if (tile >= A.tiles) return;
    TR(0);
    const int mt_run = A.mt_run;                    // 16-row tiles in this M block (<= 8)
    const int m0 = mb * mt_run * 16;
    const int bm = min(mt_run * 16, A.M - m0);
    const int p0 = tile * P2_BN;
    const int K = A.K, P = A.P;
    const int kgroups = (K + 15) / 16;

    // ---- activation staging descriptors (chunk-invariant) ----------------------------------
    // 64 rows x 16 float4 columns = 1024 slots, four per thread
    int brow[P2_NB], bcol[P2_NB], boff[P2_NB][VEC ? 1 : 4];
    bool bval[P2_NB][VEC ? 1 : 4];
#pragma unroll
    for (int i = 0; i < P2_NB; ++i) {
        const int slot = i * 256 + tid;
        brow[i] = slot >> 4;
        bcol[i] = (slot & 15) * 4;
#pragma unroll
        for (int e = 0; e < (VEC ? 1 : 4); ++e) {
            const int p = p0 + bcol[i] + e;
            bval[i][e] = p < P;
            int off = p;
            if (!VEC && IN != IN_BNBWD && A.strided && bval[i][e]) {
                const int hw = A.Ho * A.Wo;
                const int t = p / hw, rem = p - t * hw;
                const int ho = rem / A.Wo, wo = rem - ho * A.Wo;
                off = (t * A.H + 2 * ho) * A.W + 2 * wo;
            }
            boff[i][e] = bval[i][e] ? off : 0;        // clamped: always a valid address
        }
    }
    const float* xn = A.x + (size_t)n * K * (size_t)A.Pin;
    const float* an = (IN == IN_BNBWD) ? A.a + (size_t)n * K * (size_t)P : nullptr;

    float4 rb[P2_NB], rba[(IN == IN_BNBWD) ? P2_NB : 1];

    // Branch-free fetch: every load is unconditional from a clamped address; predicates are
    // applied when the registers are written to LDS (a load under a divergent branch makes the
    // compiler wait for it before the next branch and serialises the burst on memory latency).
    auto fetch = [&](int c) {
        const int k0 = c * P2_KC;
#pragma unroll
        for (int i = 0; i < P2_NB; ++i) {
            const int kc = min(k0 + brow[i], K - 1);
            const float* px = xn + (size_t)kc * (size_t)A.Pin;
            if (VEC) {
                rb[i] = *reinterpret_cast<const float4*>(px + boff[i][0]);
            } else {
                rb[i] = make_float4(px[boff[i][0]], px[boff[i][1 % (VEC ? 1 : 4)]], px[boff[i][2 % (VEC ? 1 : 4)]],
                                    px[boff[i][3 % (VEC ? 1 : 4)]]);
            }
            if (IN == IN_BNBWD) {
                const float* pa = an + (size_t)kc * (size_t)P;
                if (VEC) {
                    rba[i] = *reinterpret_cast<const float4*>(pa + boff[i][0]);
                } else {
                    rba[i] = make_float4(pa[boff[i][0]], pa[boff[i][1 % (VEC ? 1 : 4)]], pa[boff[i][2 % (VEC ? 1 : 4)]],
                                         pa[boff[i][3 % (VEC ? 1 : 4)]]);
                }
            }
        }
    };

    auto store = [&](int c, int buf) {
        const int k0 = c * P2_KC;
#pragma unroll
        for (int i = 0; i < P2_NB; ++i) {
            const int k = k0 + brow[i];
            const bool kv = k < K;
            const int kc = kv ? k : 0;
            bool ok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ok[e] = kv && bval[i][VEC ? 0 : e];
            float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
            if (IN == IN_BNBWD) {
                const float c0 = Cl[kc * 3], c1 = Cl[kc * 3 + 1], c2 = Cl[kc * 3 + 2];
                const float av[4] = {rba[i].x, rba[i].y, rba[i].z, rba[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok[e] ? fmaf(c0, v[e], fmaf(c1, av[e], c2)) : 0.f;
            } else if (IN == IN_AFFACT) {
                const float sc = Cl[kc * 2], sh = Cl[kc * 2 + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok[e] ? act_fwd(fmaf(sc, v[e], sh), A.in_act) : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok[e] ? v[e] : 0.f;
            }
            *reinterpret_cast<float4*>(&Bl[buf][brow[i] * P2_BN + bcol[i]]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    };

    // this wave's M tiles: wave and wave + 4 (clamped to an existing tile: results of a clamped
    // duplicate are simply never stored)
    const int mtl = (A.M + 15) / 16 - 1;           // last packed tile
    const float* wp0 = A.wp + ((size_t)min(mb * mt_run + wave, mtl) * kgroups) * 256 + lane * 4;
    const float* wp1 = A.wp + ((size_t)min(mb * mt_run + wave + 4, mtl) * kgroups) * 256 + lane * 4;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weight fragments: two register sets, the next chunk's set is requested before the current
    // chunk's MFMAs (L2 latency hidden under 128 MFMAs)
    float4 a0[P2_KC / 16], a1[P2_KC / 16], b0[P2_KC / 16], b1[P2_KC / 16];
    auto fetch_a = [&](int c, float4 (&d0)[P2_KC / 16], float4 (&d1)[P2_KC / 16]) {
#pragma unroll
        for (int s4 = 0; s4 < P2_KC / 16; ++s4) {
            const int sg = min(c * (P2_KC / 16) + s4, kgroups - 1);     // clamped; extra groups meet zero B rows
            d0[s4] = *reinterpret_cast<const float4*>(wp0 + (size_t)sg * 256);
            if (TWO) d1[s4] = *reinterpret_cast<const float4*>(wp1 + (size_t)sg * 256);
            else d1[s4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    auto compute = [&](int buf, const float4 (&w0)[P2_KC / 16], const float4 (&w1)[P2_KC / 16]) {
#pragma unroll
        for (int s4 = 0; s4 < P2_KC / 16; ++s4) {
            const float av0[4] = {w0[s4].x, w0[s4].y, w0[s4].z, w0[s4].w};
            const float av1[4] = {w1[s4].x, w1[s4].y, w1[s4].z, w1[s4].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 b = *reinterpret_cast<const float4*>(&Bl[buf][(s4 * 16 + 4 * q + e) * P2_BN + 4 * r]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[e], b.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[e], b.y, acc[0][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[e], b.z, acc[0][2], 0, 0, 0);
                acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[e], b.w, acc[0][3], 0, 0, 0);
                if (TWO) {      // M blocks of more than 4 tiles: second tile of this wave
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[e], b.x, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[e], b.y, acc[1][1], 0, 0, 0);
                    acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[e], b.z, acc[1][2], 0, 0, 0);
                    acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[e], b.w, acc[1][3], 0, 0, 0);
                }
            }
        }
    };

    const int nchunks = (K + P2_KC - 1) / P2_KC;
    // the first activation / weight requests go out before the coefficient table is filled, so
    // the two global round trips overlap
    fetch(0);
    fetch_a(0, a0, a1);
    if (IN != IN_RAW) {
        for (int i = tid; i < K * NC; i += 256) Cl[i] = A.cin[(size_t)n * K * NC + i];
    }
    TR(1);
    __syncthreads();            // Cl visible
    TR(2);
    store(0, 0);
    __syncthreads();
    TR(3);
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 1 < nchunks) { fetch(c + 1); fetch_a(c + 1, b0, b1); }
        compute(0, a0, a1);
        if (c + 1 < nchunks) store(c + 1, 1);
        __syncthreads();
        if (c + 1 < nchunks) {
            if (c + 2 < nchunks) { fetch(c + 2); fetch_a(c + 2, a0, a1); }
            compute(1, b0, b1);
            if (c + 2 < nchunks) store(c + 2, 0);
            __syncthreads();
        }
    }

    TR(4);
    // ------------------------------ epilogue ------------------------------
    // Phase 1 issues every global read of the epilogue (activation-derivative input, its BN
    // coefficients, the residual addend) from clamped addresses with no divergent branch around
    // them; phase 2 combines and stores.  (A load under a per-lane branch is waited for before
    // the next branch: eight rows would pay eight serial memory round trips.)
    const int pl = p0 + 4 * r;                      // this lane's first voxel
    bool pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pv[j] = pl + j < P;
    const bool has_add = EPI != EPI_STATS && A.addend != nullptr;
    const bool add_s2 = has_add && A.addend_stride == 2;
    int aoff[4];
    bool av[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = min(pl + j, P - 1);
        av[j] = has_add && pv[j];
        aoff[j] = p;
        if (add_s2) {
            const int hw = A.H * A.W;
            const int t = p / hw, rem = p - t * hw;
            const int h = rem / A.W, w = rem - h * A.W;
            const bool even = !(h & 1) && !(w & 1);
            av[j] = av[j] && even;
            aoff[j] = even ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : 0;
        }
    }
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;
    constexpr int NI = TWO ? 2 : 1;
    float xv[NI][4][4], mk[NI][4][4], adv[NI][4][4], esc[NI][4], esh[NI][4];
    bool mvv[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int lt = wave + 4 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ml = lt * 16 + 4 * q + e;
            mvv[i][e] = lt < mt_run && ml < bm;
            const size_t mrow = (size_t)n * A.M + m0 + (mvv[i][e] ? ml : 0);       // clamped to a valid row
            if (EPI == EPI_ACTBWD) {
                const float2 c2 = *reinterpret_cast<const float2*>(A.ecoef + mrow * 2);
                esc[i][e] = c2.x; esh[i][e] = c2.y;
            }
            if (EPI_HAS_X(EPI)) {
                const float* px = A.ex + mrow * (size_t)P;
                if (VEC) {
                    const float4 t4 = *reinterpret_cast<const float4*>(px + (pv[0] ? pl : 0));
                    xv[i][e][0] = t4.x; xv[i][e][1] = t4.y; xv[i][e][2] = t4.z; xv[i][e][3] = t4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[i][e][j] = px[min(pl + j, P - 1)];
                }
            }
            if (EPI == EPI_RESBWD) {
                const float* pm = A.emask + mrow * (size_t)P;
                if (VEC) {
                    const float4 t4 = *reinterpret_cast<const float4*>(pm + (pv[0] ? pl : 0));
                    mk[i][e][0] = t4.x; mk[i][e][1] = t4.y; mk[i][e][2] = t4.z; mk[i][e][3] = t4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mk[i][e][j] = pm[min(pl + j, P - 1)];
                }
            }
            if (has_add) {
                const float* pa = A.addend + mrow * (size_t)addP;
                if (VEC && !add_s2) {
                    const float4 t4 = *reinterpret_cast<const float4*>(pa + (pv[0] ? pl : 0));
                    adv[i][e][0] = t4.x; adv[i][e][1] = t4.y; adv[i][e][2] = t4.z; adv[i][e][3] = t4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) adv[i][e][j] = pa[aoff[j]];
                }
            }
        }
    }
    TR(5);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int lt = wave + 4 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ml = lt * 16 + 4 * q + e;
            const int m = m0 + ml;
            const bool mv = mvv[i][e];
            float v[4] = {acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]};
            float s1 = 0.f, s2 = 0.f;
            if (has_add) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += av[j] ? adv[i][e][j] : 0.f;
            }
            if (EPI_HAS_X(EPI)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xj = pv[j] ? xv[i][e][j] : 0.f;
                    if (EPI == EPI_RESBWD) v[j] = (pv[j] && mk[i][e][j] > 0.f) ? v[j] : 0.f;
                    else v[j] = pv[j] ? v[j] * act_bwd(fmaf(esc[i][e], xj, esh[i][e]), A.e_act) : 0.f;
                    s1 += v[j];
                    s2 = fmaf(v[j], xj, s2);
                }
            } else if (EPI == EPI_STATS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = pv[j] ? v[j] : 0.f; s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
            }
            if (mv && pv[0]) {
                float* py = A.y + ((size_t)n * A.M + m) * (size_t)P + pl;
                if (VEC) {
                    *reinterpret_cast<float4*>(py) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (pv[j]) py[j] = v[j];
                }
            }
            if (EPI != EPI_PLAIN && A.partial != nullptr) {
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if (r == 0 && mv) {
                    A.partial[(((size_t)n * A.M + m) * A.tiles + tile) * 2] = s1;
                    A.partial[(((size_t)n * A.M + m) * A.tiles + tile) * 2 + 1] = s2;
                }
            }
        }
    }
#ifdef X3D_TRACE
    if (tid == 0) {
        trv[6] = wall_clock64();
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        trv[7] = ((unsigned long long)xcc << 32) | hwid;
        const size_t id = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        if (id < 16384)
            for (int i = 0; i < 8; ++i) g_trace[id * 8 + i] = trv[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------
// Persistent, software-pipelined form of the tiled kernel for dense inputs with P % 4 == 0.
// pw2_kernel's workgroups all start together and run load -> MFMA -> store in lock step, so
// the memory phases and the MFMA phase never overlap (per-workgroup timeline:
// profiles/r01/timelines.txt).  Here a workgroup walks a strided list of work items
// (sample, 64-voxel tile, M block) as ONE flat sequence of 32-channel chunks:
//   * the global reads of chunk g+1 (activations, per-channel coefficients, packed weight
//     fragments) are issued before the MFMAs of chunk g and written to the other LDS buffer
//     after them -- one barrier per chunk, item boundaries included, so the epilogue stores of
//     item i overlap the loads of item i+1;
//   * no per-sample coefficient table in LDS: the (<= 3) coefficients of a staged channel row
//     ride along with its activation float4;
//   * work split inside the workgroup: unit = (16-row M tile, 32-voxel half tile); wave w owns
//     half (w & 1) and M tiles (w >> 1) + 2j, j < U, so 6 M tiles (96 channels) occupy all four
//     waves evenly (pw2_kernel: 75 %).
//   * item -> workgroup map keeps the M blocks of one voxel tile on one XCD (ids differ by 8).
// ---------------------------------------------------------------------------------------
constexpr int P4_BN = 64, P4_KC = 32, P4_PITCH = 72;     // pitch 72: the four k rows of a b64 fragment read hit disjoint banks

template <int IN, int EPI, int U>
__global__ __launch_bounds__(256, 2) void pw4_kernel(const PwArgs A) {
    __shared__ __attribute__((aligned(16))) float Bl[2][P4_KC * P4_PITCH];
    __shared__ float red[(EPI == EPI_PLAIN) ? 4 : 4 * U * 16 * 2];
    constexpr int NC = (IN == IN_BNBWD) ? 3 : 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int half = wave & 1, mpar = wave >> 1;
    const int K = A.K, P = A.P, M = A.M;
    const int kgroups = (K + 15) / 16, nchunks = (K + P4_KC - 1) / P4_KC;
    const int mt_run = A.mt_run, mblocks = A.mblocks;
    // item = ((voxel-tile group of 8) * mblocks + mb) * 8 + (voxel tile & 7), voxel tiles numbered over ALL samples
    // (n * tiles + tile): item % 8 -- hence the XCD of the workgroup that owns it -- is the low tile bits, so the M
    // blocks of one voxel tile share an XCD, and every workgroup has work even when a sample has < 8 tiles
    const int VT = A.N * A.tiles;
    const int items = ((VT + 7) / 8) * 8 * mblocks;
    const int mtl = (M + 15) / 16 - 1;              // last packed M tile
    const int G = gridDim.x;

    auto next_valid = [&](int it, int& n, int& tile, int& mb) {
        while (it < items) {
            const int tlo = it & 7, rest = it >> 3;
            mb = rest % mblocks;
            const int vt = (rest / mblocks) * 8 + tlo;
            if (vt < VT) {
                n = vt / A.tiles;
                tile = vt - n * A.tiles;
                break;
            }
            it += G;
        }
        return it;
    };

    // staging slots of this thread: rows (tid >> 4) and (tid >> 4) + 16 of a chunk, voxels col..col+3
    const int srow = tid >> 4, scol = (tid & 15) * 4;

    float4 rb[2], rba[(IN == IN_BNBWD) ? 2 : 1];
    float cf[2][NC];
    bool okm[2];

    auto prefetch = [&](int n, int tile, int c) {
        const int p = tile * P4_BN + scol;
        const bool pvalid = p < P;
        const int poff = pvalid ? p : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = c * P4_KC + srow + 16 * i;
            const int kc = min(k, K - 1);
            okm[i] = pvalid && k < K;
            const size_t rowi = (size_t)n * K + kc;
            rb[i] = *reinterpret_cast<const float4*>(A.x + rowi * (size_t)P + poff);
            if (IN == IN_BNBWD) rba[i] = *reinterpret_cast<const float4*>(A.a + rowi * (size_t)P + poff);
            if (IN == IN_AFFACT) {
                const float2 c2 = *reinterpret_cast<const float2*>(A.cin + rowi * 2);
                cf[i][0] = c2.x; cf[i][1] = c2.y;
            } else if (IN == IN_BNBWD) {
                const float* pc = A.cin + rowi * 3;
                cf[i][0] = pc[0]; cf[i][1] = pc[1]; cf[i][2 % NC] = pc[2];
            }
        }
    };

    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
            if (IN == IN_BNBWD) {
                const float av[4] = {rba[i].x, rba[i].y, rba[i].z, rba[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(cf[i][0], v[e], fmaf(cf[i][1], av[e], cf[i][2 % NC]));
            } else if (IN == IN_AFFACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(fmaf(cf[i][0], v[e], cf[i][1]), A.in_act);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = okm[i] ? v[e] : 0.f;
            *reinterpret_cast<float4*>(&Bl[buf][(srow + 16 * i) * P4_PITCH + scol]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    };

    auto fetch_a = [&](int mb, int c, float4 (&d)[U][2]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int mt = min(mb * mt_run + mpar + 2 * j, mtl);             // clamped: duplicates are never stored
            const float* base = A.wp + (size_t)mt * kgroups * 256 + lane * 4;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int sg = min(2 * c + s2, kgroups - 1);                 // clamped: the extra group meets zero B rows
                d[j][s2] = *reinterpret_cast<const float4*>(base + (size_t)sg * 256);
            }
        }
    };

    f32x4 acc[U][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < U; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    };

    auto compute = [&](int buf, const float4 (&a)[U][2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float2 b = *reinterpret_cast<const float2*>(&Bl[buf][(s2 * 16 + 4 * q + e) * P4_PITCH + 32 * half + 2 * r]);
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const float av = e == 0 ? a[j][s2].x : (e == 1 ? a[j][s2].y : (e == 2 ? a[j][s2].z : a[j][s2].w));
                    acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, acc[j][1], 0, 0, 0);
                }
            }
        }
    };

    const bool has_add = EPI != EPI_STATS && A.addend != nullptr;
    const bool add_s2 = has_add && A.addend_stride == 2;
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;

    auto epilogue = [&](int n, int tile, int mb) {
        const int m0 = mb * mt_run * 16;
        const int bm = min(mt_run * 16, M - m0);
        const int pl = tile * P4_BN + 32 * half + 2 * r;       // this lane's two voxels (P even: both or none valid)
        const bool pv = pl < P;
        const int pc = pv ? pl : 0;
        int aoff[2] = {pc, pc + 1};
        bool av[2] = {has_add && pv, has_add && pv};
        if (add_s2) {
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int p = pc + j2;
                const int hw = A.H * A.W;
                const int t = p / hw, rem = p - t * hw;
                const int h = rem / A.W, w = rem - h * A.W;
                const bool even = !(h & 1) && !(w & 1);
                av[j2] = av[j2] && even;
                aoff[j2] = even ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int lt = mpar + 2 * j;
            // phase 1: the four rows' reads, branch-free from clamped addresses
            float xv[4][2], mk[4][2], adv[4][2], esc[4], esh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = lt * 16 + 4 * q + e;
                const size_t mrow = (size_t)n * M + m0 + ((lt < mt_run && ml < bm) ? ml : 0);
                if (EPI == EPI_ACTBWD) {
                    const float2 c2 = *reinterpret_cast<const float2*>(A.ecoef + mrow * 2);
                    esc[e] = c2.x; esh[e] = c2.y;
                }
                if (EPI_HAS_X(EPI)) {
                    const float2 t2 = *reinterpret_cast<const float2*>(A.ex + mrow * (size_t)P + pc);
                    xv[e][0] = t2.x; xv[e][1] = t2.y;
                }
                if (EPI == EPI_RESBWD) {
                    const float2 t2 = *reinterpret_cast<const float2*>(A.emask + mrow * (size_t)P + pc);
                    mk[e][0] = t2.x; mk[e][1] = t2.y;
                }
                if (has_add) {
                    const float* pa = A.addend + mrow * (size_t)addP;
                    if (!add_s2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(pa + pc);
                        adv[e][0] = t2.x; adv[e][1] = t2.y;
                    } else {
                        adv[e][0] = pa[aoff[0]]; adv[e][1] = pa[aoff[1]];
                    }
                }
            }
            // phase 2
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = lt * 16 + 4 * q + e;
                const bool mv = lt < mt_run && ml < bm;
                float v[2] = {acc[j][0][e], acc[j][1][e]};
                float s1 = 0.f, s2 = 0.f;
                if (has_add) { v[0] += av[0] ? adv[e][0] : 0.f; v[1] += av[1] ? adv[e][1] : 0.f; }
                if (EPI_HAS_X(EPI)) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const float xj = pv ? xv[e][j2] : 0.f;
                        if (EPI == EPI_RESBWD) v[j2] = (pv && mk[e][j2] > 0.f) ? v[j2] : 0.f;
                        else v[j2] = pv ? v[j2] * act_bwd(fmaf(esc[e], xj, esh[e]), A.e_act) : 0.f;
                        s1 += v[j2];
                        s2 = fmaf(v[j2], xj, s2);
                    }
                } else if (EPI == EPI_STATS) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) { v[j2] = pv ? v[j2] : 0.f; s1 += v[j2]; s2 = fmaf(v[j2], v[j2], s2); }
                }
                if (mv && pv)
                    *reinterpret_cast<float2*>(A.y + ((size_t)n * M + m0 + ml) * (size_t)P + pl) = make_float2(v[0], v[1]);
                if (EPI != EPI_PLAIN) {
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    if (r == 0) {
                        red[((wave * U + j) * 16 + 4 * q + e) * 2] = mv ? s1 : 0.f;
                        red[((wave * U + j) * 16 + 4 * q + e) * 2 + 1] = mv ? s2 : 0.f;
                    }
                }
            }
        }
        if (EPI != EPI_PLAIN && A.partial != nullptr) {
            __syncthreads();
            // one partial per (row, 64-voxel tile): the two half-tile waves of a row are summed here
            for (int idx = tid; idx < bm * 2; idx += 256) {
                const int ml = idx >> 1, which = idx & 1;
                const int lt = ml >> 4, wv = (lt & 1) * 2, j = lt >> 1;
                const float s = red[(((wv)*U + j) * 16 + (ml & 15)) * 2 + which] +
                                red[(((wv + 1) * U + j) * 16 + (ml & 15)) * 2 + which];
                A.partial[(((size_t)n * M + (m0 + ml)) * A.tiles + tile) * 2 + which] = s;
            }
        }
    };

    // ---------------------------------- flat pipeline ----------------------------------
    int n = 0, tile = 0, mb = 0;
    int it = next_valid(blockIdx.x, n, tile, mb);
    if (it >= items) return;
#ifdef X3D_TRACE
    unsigned long long trv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int tri = 1;
    trv[0] = wall_clock64();
#define TRN() do { if (tri < 7) trv[tri++] = wall_clock64(); } while (0)
#else
#define TRN() do { } while (0)
#endif
    int pn = n, ptile = tile, pmb = mb, pit = it, pc_ = 0;          // prefetch cursor: chunk pc_ of item pit
    auto advance = [&]() {
        if (++pc_ == nchunks) { pc_ = 0; pit = next_valid(pit + G, pn, ptile, pmb); }
    };
    float4 a0[U][2], a1[U][2];
    prefetch(pn, ptile, 0);
    fetch_a(pmb, 0, a0);
    advance();
    zero_acc();
    int c = 0;
    for (;;) {
        store(0);
        __syncthreads();
        bool more = pit < items;
        if (more) { prefetch(pn, ptile, pc_); fetch_a(pmb, pc_, a1); advance(); }
        compute(0, a0);
        if (++c == nchunks) { TRN(); epilogue(n, tile, mb); TRN(); zero_acc(); c = 0; it = next_valid(it + G, n, tile, mb); }
        if (!more) break;
        store(1);
        __syncthreads();
        more = pit < items;
        if (more) { prefetch(pn, ptile, pc_); fetch_a(pmb, pc_, a0); advance(); }
        compute(1, a1);
        if (++c == nchunks) { TRN(); epilogue(n, tile, mb); TRN(); zero_acc(); c = 0; it = next_valid(it + G, n, tile, mb); }
        if (!more) break;
    }
#ifdef X3D_TRACE
    if (tid == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        trv[7] = ((unsigned long long)xcc << 32) | hwid;
        if (blockIdx.x < 16384)
            for (int i = 0; i < 8; ++i) g_trace[(size_t)blockIdx.x * 8 + i] = trv[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------
// Split-bf16 data-gradient kernel (backward only): pw4_kernel's persistent pipeline with the GEMM
// on the bf16 matrix rate.  dX = W^T dY with dY = k0*g + k1*a + k2 (BN backward) staged per voxel:
//   * thread (voxel v = tid & 63, k octet = tid >> 6) loads its 8 channels of g and a (eight dword
//     loads per tensor, each coalesced over the 64 voxels of the tile), combines, splits the value
//     into hi = bf16(v), lo = bf16(v - hi) and writes ONE 16-byte row fragment per plane into an LDS
//     image [voxel][32 channels] -- exactly the B operand of v_mfma_f32_16x16x32_bf16 (lane = voxel,
//     8 consecutive k), so the transposition costs nothing;
//   * the A operand comes pre-split and pre-packed (x3d_pw_pack, transposed image);
//   * each 16x16x32 tile product is hi*hi + hi*lo + lo*hi in fp32: 3 MFMAs of 16 cycles instead
//     of 8 fp32 MFMAs of 32.
// Error ~2^-16 per product.  The backward pass is linear in dY with the ReLU masks fixed by the
// (exact fp32) forward, so this does not amplify: gradients computed this way sit on the fp32
// noise floor of the reference (tests/exp_split_precision.py, mode "bwd"); the forward GEMMs stay
// fp32 because there the same split moves the logits by 2.6e-3.
// Voxel v of a 32-voxel half tile lives in LDS row (v >> 1) + 16 (v & 1): the two 16-column MFMA
// tiles of a wave then read consecutive rows (conflict-free) and each lane owns voxels 2r, 2r+1.
// ---------------------------------------------------------------------------------------
constexpr int P5_BN = 64, P5_KC = 32, P5_LD = 40;       // bf16 elements per LDS row (80 B)

template <int EPI, int U>
__global__ __launch_bounds__(256, 2) void pw5_kernel(const PwArgs A) {
    __shared__ __attribute__((aligned(16))) __bf16 Bh[2][P5_BN * P5_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Blo[2][P5_BN * P5_LD];
    __shared__ float red[(EPI == EPI_PLAIN) ? 4 : 4 * U * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int half = wave & 1, mpar = wave >> 1;
    const int K = A.K, P = A.P, M = A.M;
    const int kg16 = (K + 15) / 16, kg32 = (K + 31) / 32, nchunks = kg32;
    const int mt_run = A.mt_run, mblocks = A.mblocks;
    // item = ((voxel-tile group of 8) * mblocks + mb) * 8 + (voxel tile & 7), voxel tiles numbered over ALL samples
    // (n * tiles + tile): item % 8 -- hence the XCD of the workgroup that owns it -- is the low tile bits, so the M
    // blocks of one voxel tile share an XCD, and every workgroup has work even when a sample has < 8 tiles
    const int VT = A.N * A.tiles;
    const int items = ((VT + 7) / 8) * 8 * mblocks;
    const int mtiles = (M + 15) / 16, mtl = mtiles - 1;
    const int G = gridDim.x;
    // split-bf16 planes sit behind the fp32 image of the transposed pack
    const __bf16* wqh = reinterpret_cast<const __bf16*>(A.wp + (size_t)mtiles * kg16 * 256);
    const __bf16* wql = wqh + (size_t)mtiles * kg32 * 512;

    auto next_valid = [&](int it, int& n, int& tile, int& mb) {
        while (it < items) {
            const int tlo = it & 7, rest = it >> 3;
            mb = rest % mblocks;
            const int vt = (rest / mblocks) * 8 + tlo;
            if (vt < VT) {
                n = vt / A.tiles;
                tile = vt - n * A.tiles;
                break;
            }
            it += G;
        }
        return it;
    };

    // staging role: voxel `lane` of the tile, channels 8 * wave .. 8 * wave + 7 of the chunk
    const int srow = (lane & 31) >> 1, sodd = lane & 1, shalf = lane >> 5;
    const int lrow = shalf * 32 + sodd * 16 + srow;                 // LDS row of this thread's voxel
    float rg[8], ra[8], cf[8][3];
    bool okv;
    int kbase;

    auto prefetch = [&](int n, int tile, int c) {
        const int p = tile * P5_BN + lane;
        okv = p < P;
        const int pc = okv ? p : 0;
        kbase = c * P5_KC + 8 * wave;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kc = min(kbase + j, K - 1);                    // wave-uniform row
            const size_t rowi = (size_t)n * K + kc;
            rg[j] = A.x[rowi * (size_t)P + pc];
            ra[j] = A.a[rowi * (size_t)P + pc];
            const float* pc3 = A.cin + rowi * 3;
            cf[j][0] = pc3[0]; cf[j][1] = pc3[1]; cf[j][2] = pc3[2];
        }
    };

    auto store = [&](int buf) {
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = fmaf(cf[j][0], rg[j], fmaf(cf[j][1], ra[j], cf[j][2]));
            v = (okv && kbase + j < K) ? v : 0.f;
            const __bf16 h = (__bf16)v;
            hi[j] = h;
            lo[j] = (__bf16)(v - (float)h);
        }
        *reinterpret_cast<bf16x8*>(&Bh[buf][lrow * P5_LD + 8 * wave]) = hi;
        *reinterpret_cast<bf16x8*>(&Blo[buf][lrow * P5_LD + 8 * wave]) = lo;
    };

    auto fetch_a = [&](int mb, int c, bf16x8 (&dh)[U], bf16x8 (&dl)[U]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int mt = min(mb * mt_run + mpar + 2 * j, mtl);             // clamped: duplicates are never stored
            const size_t off = (((size_t)mt * kg32 + c) * 64 + lane) * 8;
            dh[j] = *reinterpret_cast<const bf16x8*>(wqh + off);
            dl[j] = *reinterpret_cast<const bf16x8*>(wql + off);
        }
    };

    f32x4 acc[U][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < U; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    };

    auto compute = [&](int buf, const bf16x8 (&ah)[U], const bf16x8 (&al)[U]) {
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int off = (half * 32 + h2 * 16 + r) * P5_LD + 8 * q;      // column r of tile h2 = voxel 32 half + 2 r + h2
            bh[h2] = *reinterpret_cast<const bf16x8*>(&Bh[buf][off]);
            bl[h2] = *reinterpret_cast<const bf16x8*>(&Blo[buf][off]);
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j], bh[h2], acc[j][h2], 0, 0, 0);
                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bl[h2], acc[j][h2], 0, 0, 0);
                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[j], bh[h2], acc[j][h2], 0, 0, 0);
            }
    };

    const bool has_add = A.addend != nullptr;
    const bool add_s2 = has_add && A.addend_stride == 2;
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;

    // identical to pw4_kernel's epilogue (lane: rows 4q + e of each unit, voxels 32 half + 2 r, + 1)
    auto epilogue = [&](int n, int tile, int mb) {
        const int m0 = mb * mt_run * 16;
        const int bm = min(mt_run * 16, M - m0);
        const int pl = tile * P5_BN + 32 * half + 2 * r;
        const bool pv = pl < P;
        const int pc = pv ? pl : 0;
        int aoff[2] = {pc, pc + 1};
        bool av[2] = {has_add && pv, has_add && pv};
        if (add_s2) {
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int p = pc + j2;
                const int hw = A.H * A.W;
                const int t = p / hw, rem = p - t * hw;
                const int h = rem / A.W, w = rem - h * A.W;
                const bool even = !(h & 1) && !(w & 1);
                av[j2] = av[j2] && even;
                aoff[j2] = even ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int lt = mpar + 2 * j;
            float xv[4][2], mk[4][2], adv[4][2], esc[4], esh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = lt * 16 + 4 * q + e;
                const size_t mrow = (size_t)n * M + m0 + ((lt < mt_run && ml < bm) ? ml : 0);
                if (EPI == EPI_ACTBWD) {
                    const float2 c2 = *reinterpret_cast<const float2*>(A.ecoef + mrow * 2);
                    esc[e] = c2.x; esh[e] = c2.y;
                }
                if (EPI_HAS_X(EPI)) {
                    const float2 t2 = *reinterpret_cast<const float2*>(A.ex + mrow * (size_t)P + pc);
                    xv[e][0] = t2.x; xv[e][1] = t2.y;
                }
                if (EPI == EPI_RESBWD) {
                    const float2 t2 = *reinterpret_cast<const float2*>(A.emask + mrow * (size_t)P + pc);
                    mk[e][0] = t2.x; mk[e][1] = t2.y;
                }
                if (has_add) {
                    const float* pa = A.addend + mrow * (size_t)addP;
                    if (!add_s2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(pa + pc);
                        adv[e][0] = t2.x; adv[e][1] = t2.y;
                    } else {
                        adv[e][0] = pa[aoff[0]]; adv[e][1] = pa[aoff[1]];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = lt * 16 + 4 * q + e;
                const bool mv = lt < mt_run && ml < bm;
                float v[2] = {acc[j][0][e], acc[j][1][e]};
                float s1 = 0.f, s2 = 0.f;
                if (has_add) { v[0] += av[0] ? adv[e][0] : 0.f; v[1] += av[1] ? adv[e][1] : 0.f; }
                if (EPI_HAS_X(EPI)) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const float xj = pv ? xv[e][j2] : 0.f;
                        if (EPI == EPI_RESBWD) v[j2] = (pv && mk[e][j2] > 0.f) ? v[j2] : 0.f;
                        else v[j2] = pv ? v[j2] * act_bwd(fmaf(esc[e], xj, esh[e]), A.e_act) : 0.f;
                        s1 += v[j2];
                        s2 = fmaf(v[j2], xj, s2);
                    }
                }
                if (mv && pv)
                    *reinterpret_cast<float2*>(A.y + ((size_t)n * M + m0 + ml) * (size_t)P + pl) = make_float2(v[0], v[1]);
                if (EPI != EPI_PLAIN) {
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    if (r == 0) {
                        red[((wave * U + j) * 16 + 4 * q + e) * 2] = mv ? s1 : 0.f;
                        red[((wave * U + j) * 16 + 4 * q + e) * 2 + 1] = mv ? s2 : 0.f;
                    }
                }
            }
        }
        if (EPI != EPI_PLAIN && A.partial != nullptr) {
            __syncthreads();
            for (int idx = tid; idx < bm * 2; idx += 256) {
                const int ml = idx >> 1, which = idx & 1;
                const int lt = ml >> 4, wv = (lt & 1) * 2, j = lt >> 1;
                const float s = red[(((wv)*U + j) * 16 + (ml & 15)) * 2 + which] +
                                red[(((wv + 1) * U + j) * 16 + (ml & 15)) * 2 + which];
                A.partial[(((size_t)n * M + (m0 + ml)) * A.tiles + tile) * 2 + which] = s;
            }
        }
    };

    // ---------------------------------- flat pipeline (as pw4_kernel) ----------------------------------
    int n = 0, tile = 0, mb = 0;
    int it = next_valid(blockIdx.x, n, tile, mb);
    if (it >= items) return;
    int pn = n, ptile = tile, pmb = mb, pit = it, pc_ = 0;
    auto advance = [&]() {
        if (++pc_ == nchunks) { pc_ = 0; pit = next_valid(pit + G, pn, ptile, pmb); }
    };
    bf16x8 a0h[U], a0l[U], a1h[U], a1l[U];
    prefetch(pn, ptile, 0);
    fetch_a(pmb, 0, a0h, a0l);
    advance();
    zero_acc();
    int c = 0;
    for (;;) {
        store(0);
        __syncthreads();
        bool more = pit < items;
        if (more) { prefetch(pn, ptile, pc_); fetch_a(pmb, pc_, a1h, a1l); advance(); }
        compute(0, a0h, a0l);
        if (++c == nchunks) { epilogue(n, tile, mb); zero_acc(); c = 0; it = next_valid(it + G, n, tile, mb); }
        if (!more) break;
        store(1);
        __syncthreads();
        more = pit < items;
        if (more) { prefetch(pn, ptile, pc_); fetch_a(pmb, pc_, a0h, a0l); advance(); }
        compute(1, a1h, a1l);
        if (++c == nchunks) { epilogue(n, tile, mb); zero_acc(); c = 0; it = next_valid(it + G, n, tile, mb); }
        if (!more) break;
    }
}

// One decision function for kernel variant and tile count (the caller sizes `partial` with it).
// variant: 0 = streaming NT=4, 1 = streaming NT=1, 2 = LDS-tiled (pw2)
static void pw_plan(int N, int K, int M, int P, bool dense, int* variant, int* tiles, int* mblocks, int* mt_run) {
    const int mtiles = cdiv(M, 16);
    const bool no_persist = x3d_opt(X3D_OPT_PW_NO_PERSIST) != 0;         // A/B knob (tests, tools/microbench.py); read per call
    if (K >= 64 && M >= 96 && dense && (P % 4 == 0) && !no_persist) {
        // persistent pipelined kernel: units of (M tile, half voxel tile), U = ceil(mt_run / 2) per wave;
        // fewest M blocks (each restages the activation tile) unless one more block removes a whole unit row
        *variant = 3;
        int best_mb = cdiv(mtiles, 8), best_cost = 1 << 30;
        for (int mbk = cdiv(mtiles, 8); mbk <= cdiv(mtiles, 8) + 1; ++mbk) {
            const int run = cdiv(mtiles, mbk);
            const int cost = mbk * cdiv(run, 2);
            if (cost < best_cost) { best_cost = cost; best_mb = mbk; }
        }
        *mblocks = best_mb;
        *mt_run = cdiv(mtiles, best_mb);
        *tiles = cdiv(P, P4_BN);
        return;
    }
    if (K >= 64 && M >= 96) {
        *variant = 2;
        // waves own tiles {w, w+4}: a block of 5..8 tiles costs two tile-times, 1..4 tiles one;
        // pick the blocking with the smaller (blocks x tile-times); ties -> fewer blocks (the
        // activation tile is staged once per block)
        const int b8 = cdiv(mtiles, 8), b4 = cdiv(mtiles, 4);
        const int r8 = cdiv(mtiles, b8);
        const int cost8 = b8 * (r8 > 4 ? 2 : 1), cost4 = b4;
        *mblocks = (cost4 < cost8) ? b4 : b8;
        *mt_run = cdiv(mtiles, *mblocks);
        *tiles = cdiv(P, P2_BN);
        return;
    }
    *mblocks = cdiv(mtiles, 4);
    *mt_run = cdiv(mtiles, *mblocks);
    int v = (dense && (P % 4 == 0)) ? 4 : 1;
    if (v == 4) {
        const long long wgs = (long long)cdiv(P, 256) * N * (*mblocks);
        // small-P layers: 16 voxels per wave (NT = 1) gives 4x the waves -- but the float4 form moves the same bytes with
        // a quarter of the memory instructions and wins as soon as its workgroups cover the CUs (stage 2 at the base
        // shape: 392 / 784 workgroups, +1.4 % on the step), unless the last 256-voxel tile of a sample is mostly empty
        const long long nt4_min = x3d_opt(X3D_OPT_PW_NT4_MIN);
        const bool full_tiles = (long long)P * 10 >= (long long)cdiv(P, 256) * 256 * 9;
        if (wgs < nt4_min || (wgs < 1024 && !full_tiles)) v = 1;
    }
    *variant = v == 4 ? 0 : 1;
    *tiles = cdiv(P, 64 * v);
}

template <int IN, int EPI>
int launch_pw(PwArgs& A, hipStream_t s) {
    int variant;
    const bool dense = !A.strided && (A.Pin % 4 == 0);
    pw_plan(A.N, A.K, A.M, A.P, dense, &variant, &A.tiles, &A.mblocks, &A.mt_run);
    dim3 grid(cdiv(A.tiles, 8) * 8 * A.mblocks, A.N), block(256);
    // mixed storage: only the packed streaming kernel (pw3) of this file reads / writes bf16 tensors
    const bool mx = A.x_bf || A.y_bf || A.ex_bf;
    if (mx && variant >= 2 && A.wp != nullptr) {
        // large-channel layers outside the whole-K kernels of pw6.hip -- voxel count not a multiple of 4 (odd clip sizes), or
        // K beyond their LDS tile (X3D-XL: 630 channels): the streaming kernel on the same 64-voxel tiles (P2_BN = P4_BN = 64),
        // as in the "no packed weights" case below
        variant = 1;
        const int mtiles = cdiv(A.M, 16);
        A.mblocks = cdiv(mtiles, 4);
        A.mt_run = cdiv(mtiles, A.mblocks);
        grid = dim3(cdiv(A.tiles, 8) * 8 * A.mblocks, A.N);
    }
    if (mx && !(variant <= 1 && A.wp != nullptr && A.K <= PW_MAXK)) {
        x3d_set_error("pw: no mixed-storage kernel for K=%d M=%d P=%d (packed=%d)", A.K, A.M, A.P, A.wp != nullptr);
        return X3D_EINVAL;
    }
    if (variant == 3 && A.wp != nullptr) {
        const int items = cdiv(A.tiles * A.N, 8) * 8 * A.mblocks;
        const int pg_max = x3d_opt(X3D_OPT_PW_PGRID);
        dim3 pgrid(min(items, pg_max));         // two resident workgroups per CU walk the item list
        const int U = cdiv(A.mt_run, 2);
        if (IN == IN_BNBWD && EPI != EPI_STATS && !x3d_opt(X3D_OPT_DGRAD_F32) && x3d_opt(X3D_OPT_BWD_TERMS) == 2) {
            // backward-data, two-term mode only: split-bf16 MFMA, hi + lo (the transposed pack carries the bf16 planes).  With
            // three-term operands (the default) the shapes that reach this point -- K beyond pw7's LDS tile: X3D-XL's 630
            // channels -- run on the exact fp32-MFMA pw4 below
            constexpr int E5 = (EPI == EPI_STATS) ? EPI_PLAIN : EPI;
            x3d_note_kernel("pw5_kernel");
            if (U <= 2) hipLaunchKernelGGL((pw5_kernel<E5, 2>), pgrid, block, 0, s, A);
            else if (U == 3) hipLaunchKernelGGL((pw5_kernel<E5, 3>), pgrid, block, 0, s, A);
            else hipLaunchKernelGGL((pw5_kernel<E5, 4>), pgrid, block, 0, s, A);
            X3D_LAUNCH_CHECK();
            return X3D_OK;
        }
        x3d_note_kernel("pw4_kernel");
        if (U <= 2) hipLaunchKernelGGL((pw4_kernel<IN, EPI, 2>), pgrid, block, 0, s, A);
        else if (U == 3) hipLaunchKernelGGL((pw4_kernel<IN, EPI, 3>), pgrid, block, 0, s, A);
        else hipLaunchKernelGGL((pw4_kernel<IN, EPI, 4>), pgrid, block, 0, s, A);
        X3D_LAUNCH_CHECK();
        return X3D_OK;
    }
    if (variant >= 2 && A.wp == nullptr) {      // no packed weights: streaming kernel on the same 64-voxel tiles
        variant = 1;
        const int mtiles = cdiv(A.M, 16);
        A.mblocks = cdiv(mtiles, 4);
        A.mt_run = cdiv(mtiles, A.mblocks);
        grid = dim3(cdiv(A.tiles, 8) * 8 * A.mblocks, A.N);
    }
    if (variant == 2) {
        if (A.K > PW_MAXK) { x3d_set_error("pw: K=%d exceeds the coefficient table (%d)", A.K, PW_MAXK); return X3D_EINVAL; }
        const bool vec = dense && (A.P % 4 == 0);
        x3d_note_kernel("pw2_kernel");
        if (A.mt_run > 4) {
            if (vec) hipLaunchKernelGGL((pw2_kernel<IN, EPI, true, true>), grid, block, 0, s, A);
            else hipLaunchKernelGGL((pw2_kernel<IN, EPI, false, true>), grid, block, 0, s, A);
        } else {
            if (vec) hipLaunchKernelGGL((pw2_kernel<IN, EPI, true, false>), grid, block, 0, s, A);
            else hipLaunchKernelGGL((pw2_kernel<IN, EPI, false, false>), grid, block, 0, s, A);
        }
    } else if (A.wp != nullptr && A.K <= PW_MAXK) {
#define PW3_GO(MT_, NT_)                                                                                    \
    do {                                                                                                    \
        if (mx) hipLaunchKernelGGL((pw3_kernel<MT_, NT_, IN, EPI, true>), grid, block, 0, s, A);             \
        else hipLaunchKernelGGL((pw3_kernel<MT_, NT_, IN, EPI, false>), grid, block, 0, s, A);               \
    } while (0)
        x3d_note_kernel("pw3_kernel");
        if (A.mt_run <= 2) {
            if (variant == 0) PW3_GO(2, 4); else PW3_GO(2, 1);
        } else {
            if (variant == 0) PW3_GO(4, 4); else PW3_GO(4, 1);
        }
#undef PW3_GO
    } else if (A.mt_run <= 2) {
        x3d_note_kernel("pw_kernel");
        if (variant == 0) hipLaunchKernelGGL((pw_kernel<2, 4, IN, EPI>), grid, block, 0, s, A);
        else hipLaunchKernelGGL((pw_kernel<2, 1, IN, EPI>), grid, block, 0, s, A);
    } else {
        x3d_note_kernel("pw_kernel");
        if (variant == 0) hipLaunchKernelGGL((pw_kernel<4, 4, IN, EPI>), grid, block, 0, s, A);
        else hipLaunchKernelGGL((pw_kernel<4, 1, IN, EPI>), grid, block, 0, s, A);
    }
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

// ---------------------------------------------------------------------------------------
// Backward-weight: dW[co][ci] = sum_{n,p} dY[co][p] * in[ci][p]   (K index = voxel).
// No LDS staging: with the voxel index on the MFMA K dimension, lane (r = lane & 15,
// q = lane >> 4) of a 16x16x4 MFMA needs channel row r at voxels 4q..4q+3 of a 16-voxel step,
// which is exactly one float4 per lane straight from HBM (64 B contiguous per row and step;
// a wave walks 128 consecutive voxels = whole 512-B row runs).  Every wave owns a private set
// of 128-voxel units and the whole (co-block x ci-block) of dW in its accumulators; the BN
// backward combine (dY = cb0*g + cb1*a + cb2) and the forward prologue (act(pre*x+pre)) are
// applied in registers.  The next step's loads are issued before the current step's MFMAs.
// Waves are summed through LDS in fixed order; one partial per workgroup goes to HBM.
// ---------------------------------------------------------------------------------------
constexpr int WG_UNIT = 128;   // voxels per work unit (8 MFMA steps of 16)

struct WgArgs {
    const float* g; const float* a; const float* cb;      // [N][Co][P], [N][Co][P], [N][Co][3]
    const float* x; const float* pre; int pre_act;        // [N][Ci][Pin], [N][Ci][2] or NULL
    float* wpartial;                                       // [groups][Co][Ci]
    int N, Ci, Co, P; long long Pin;
    int strided, T, H, W, Ho, Wo;
    int groups, units_per_sample, cob, cib, ct_run, it_run;
    int ga_bf, x_bf;                                       // mixed-storage builds (MX) of the split-bf16 kernels: g and a / x are bf16 arrays
};

template <int CT, int IT, bool VEC, bool MX>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(const WgArgs A) {
    const int ga_bf = MX ? A.ga_bf : 0, x_bf = MX ? A.x_bf : 0;
    __shared__ float red[4 * IT * 4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int blk = blockIdx.y;
    const int co0 = (blk / A.cib) * (A.ct_run * 16), ci0 = (blk % A.cib) * (A.it_run * 16);
    const int ct_run = A.ct_run, it_run = A.it_run;
    const int P = A.P;

    f32x4 acc[CT][IT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // rows of this lane
    int co[CT], ci[IT];
    bool cov[CT], civ[IT];
#pragma unroll
    for (int i = 0; i < CT; ++i) { co[i] = co0 + 16 * i + r; cov[i] = (i < ct_run) && co[i] < A.Co; if (!cov[i]) co[i] = 0; }
#pragma unroll
    for (int j = 0; j < IT; ++j) { ci[j] = ci0 + 16 * j + r; civ[j] = (j < it_run) && ci[j] < A.Ci; if (!civ[j]) ci[j] = 0; }

    const int total_units = A.N * A.units_per_sample;
    const int gw = blockIdx.x * 4 + wave, GW = A.groups * 4;

    float4 ng[CT], na[CT], nx[IT];     // next step's raw loads

    // Branch-free: unconditional loads from clamped (always valid) addresses; rows / voxels that
    // do not exist are zeroed when the fragments are formed (a load under a divergent branch is
    // waited for before the next branch, which would serialise the burst on memory latency).
    auto issue = [&](int n, int p) {   // p = first voxel of this lane's float4
        const int pc = VEC ? min(p, P - 4) : p;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            if (i < ct_run) {
                const size_t rowb = ((size_t)n * A.Co + co[i]) * (size_t)P;
                if (VEC) {
                    ng[i] = ldx4(A.g, rowb + pc, ga_bf);
                    na[i] = ldx4(A.a, rowb + pc, ga_bf);
                } else {
                    float tg[4], ta[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int pe = min(p + e, P - 1);
                        tg[e] = ldx1(A.g, rowb + pe, ga_bf);
                        ta[e] = ldx1(A.a, rowb + pe, ga_bf);
                    }
                    ng[i] = make_float4(tg[0], tg[1], tg[2], tg[3]);
                    na[i] = make_float4(ta[0], ta[1], ta[2], ta[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            if (j < it_run) {
                const size_t xrow = ((size_t)n * A.Ci + ci[j]) * (size_t)A.Pin;
                if (VEC) {
                    nx[j] = ldx4(A.x, xrow + pc, x_bf);
                } else {
                    float tx[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int pp = min(p + e, P - 1);
                        size_t off = pp;
                        if (A.strided) {
                            const int hw = A.Ho * A.Wo;
                            const int t = pp / hw, rem = pp - t * hw;
                            const int ho = rem / A.Wo, wo = rem - ho * A.Wo;
                            off = (size_t)(t * A.H + 2 * ho) * A.W + 2 * wo;
                        }
                        tx[e] = ldx1(A.x, xrow + off, x_bf);
                    }
                    nx[j] = make_float4(tx[0], tx[1], tx[2], tx[3]);
                }
            }
        }
    };

    for (int u = gw; u < total_units; u += GW) {
        const int n = u / A.units_per_sample, pu = (u - n * A.units_per_sample) * WG_UNIT;
        // per-sample row constants
        float k0[CT], k1[CT], k2[CT], sc[IT], sh[IT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const float* cb = A.cb + ((size_t)n * A.Co + co[i]) * 3;
            k0[i] = cb[0]; k1[i] = cb[1]; k2[i] = cb[2];
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            sc[j] = 1.f; sh[j] = 0.f;
            if (A.pre != nullptr) { sc[j] = A.pre[((size_t)n * A.Ci + ci[j]) * 2]; sh[j] = A.pre[((size_t)n * A.Ci + ci[j]) * 2 + 1]; }
        }
        issue(n, pu + 4 * q);
#pragma unroll 1
        for (int st = 0; st < WG_UNIT / 16; ++st) {
            const int p = pu + st * 16 + 4 * q;
            float4 dy[CT], xin[IT];
            bool pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = (p + e) < P;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                dy[i].x = (cov[i] && pv[0]) ? fmaf(k0[i], ng[i].x, fmaf(k1[i], na[i].x, k2[i])) : 0.f;
                dy[i].y = (cov[i] && pv[1]) ? fmaf(k0[i], ng[i].y, fmaf(k1[i], na[i].y, k2[i])) : 0.f;
                dy[i].z = (cov[i] && pv[2]) ? fmaf(k0[i], ng[i].z, fmaf(k1[i], na[i].z, k2[i])) : 0.f;
                dy[i].w = (cov[i] && pv[3]) ? fmaf(k0[i], ng[i].w, fmaf(k1[i], na[i].w, k2[i])) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                if (A.pre != nullptr) {
                    xin[j].x = (civ[j] && pv[0]) ? act_fwd(fmaf(sc[j], nx[j].x, sh[j]), A.pre_act) : 0.f;
                    xin[j].y = (civ[j] && pv[1]) ? act_fwd(fmaf(sc[j], nx[j].y, sh[j]), A.pre_act) : 0.f;
                    xin[j].z = (civ[j] && pv[2]) ? act_fwd(fmaf(sc[j], nx[j].z, sh[j]), A.pre_act) : 0.f;
                    xin[j].w = (civ[j] && pv[3]) ? act_fwd(fmaf(sc[j], nx[j].w, sh[j]), A.pre_act) : 0.f;
                } else {
                    xin[j].x = (civ[j] && pv[0]) ? nx[j].x : 0.f;
                    xin[j].y = (civ[j] && pv[1]) ? nx[j].y : 0.f;
                    xin[j].z = (civ[j] && pv[2]) ? nx[j].z : 0.f;
                    xin[j].w = (civ[j] && pv[3]) ? nx[j].w : 0.f;
                }
            }
            if (st + 1 < WG_UNIT / 16) issue(n, p + 16);      // prefetch the next step
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                if (i < ct_run) {
#pragma unroll
                    for (int j = 0; j < IT; ++j) {
                        if (j < it_run) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[i].x, xin[j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[i].y, xin[j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[i].z, xin[j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[i].w, xin[j].w, acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // sum the 4 waves (fixed order) and write this workgroup's partial.
    // D[i = co][j = ci]: lane (q, r), reg e -> co = tile*16 + 4q + e, ci = tile*16 + r
    float* out = A.wpartial + (size_t)blockIdx.x * A.Co * A.Ci;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        if (i < ct_run) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < IT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[((wave * IT + j) * 4 + e) * 64 + lane] = acc[i][j][e];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int j = 0; j < IT; ++j) {
                    if (j < it_run) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float s = (red[((0 * IT + j) * 4 + e) * 64 + lane] + red[((1 * IT + j) * 4 + e) * 64 + lane]) +
                                            (red[((2 * IT + j) * 4 + e) * 64 + lane] + red[((3 * IT + j) * 4 + e) * 64 + lane]);
                            const int oc = co0 + 16 * i + 4 * q + e, ic = ci0 + 16 * j + r;
                            if (oc < A.Co && ic < A.Ci) out[(size_t)oc * A.Ci + ic] = s;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Tiled backward-weight for the large-C layers (Co >= 64 and Ci >= 48, dense, P % 4 == 0).
// Workgroup = one (128 co x 64 ci) block of dW and a strided set of 64-voxel chunks.  Per chunk
// dY[128][64] (= cb0*g + cb1*a + cb2) and in[64][64] (prologue applied) are staged ONCE into
// LDS ([rows][68]); the four waves own 2 co-tiles x 4 ci-tiles each (32 accumulator registers)
// and read fragments with ds_read_b128 (voxel index on the MFMA K dimension, permuted
// identically on both operands).  Loads for chunk c+1 are in flight during the MFMAs of chunk c
// (registers), every load is unconditional from a clamped address.
// ---------------------------------------------------------------------------------------
constexpr int W2_CO = 128, W2_CI = 64, W2_PT = 64, W2_LD = 68;
constexpr int W2_ND = W2_CO * W2_PT / 4 / 256;     // dY float4 slots per thread (8)
constexpr int W2_NX = W2_CI * W2_PT / 4 / 256;     // in float4 slots per thread (4)

__global__ __launch_bounds__(256, 2) void pw_wgrad2_kernel(const WgArgs A) {
    __shared__ __attribute__((aligned(16))) float Dl[W2_CO * W2_LD];
    __shared__ __attribute__((aligned(16))) float Xl[W2_CI * W2_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int blk = blockIdx.y;
    const int co0 = (blk / A.cib) * W2_CO, ci0 = (blk % A.cib) * W2_CI;
    const int P = A.P;
    const int cps = (P + W2_PT - 1) / W2_PT;            // chunks per sample
    const int total = A.N * cps;

    // staging slots: row-major, 16 lanes x 16 B = 256 B contiguous per row
    const int c4 = (tid & 15) * 4;
    int drow[W2_ND], xrow_[W2_NX];
#pragma unroll
    for (int i = 0; i < W2_ND; ++i) drow[i] = i * 16 + (tid >> 4);
#pragma unroll
    for (int i = 0; i < W2_NX; ++i) xrow_[i] = i * 16 + (tid >> 4);

    float4 rg[W2_ND], ra[W2_ND], rx[W2_NX];
    float k0[W2_ND], k1[W2_ND], k2[W2_ND], sc[W2_NX], sh[W2_NX];
#pragma unroll
    for (int i = 0; i < W2_NX; ++i) { sc[i] = 1.f; sh[i] = 0.f; }

    auto fetch = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * W2_PT;
        const int pc = min(pt + c4, P - 4);
#pragma unroll
        for (int i = 0; i < W2_ND; ++i) {
            const int co = min(co0 + drow[i], A.Co - 1);
            const size_t base = ((size_t)n * A.Co + co) * (size_t)P + pc;
            rg[i] = *reinterpret_cast<const float4*>(A.g + base);
            ra[i] = *reinterpret_cast<const float4*>(A.a + base);
            const float* cb = A.cb + ((size_t)n * A.Co + co) * 3;
            k0[i] = cb[0]; k1[i] = cb[1]; k2[i] = cb[2];
        }
#pragma unroll
        for (int i = 0; i < W2_NX; ++i) {
            const int ci = min(ci0 + xrow_[i], A.Ci - 1);
            rx[i] = *reinterpret_cast<const float4*>(A.x + ((size_t)n * A.Ci + ci) * (size_t)A.Pin + pc);
            if (A.pre != nullptr) { sc[i] = A.pre[((size_t)n * A.Ci + ci) * 2]; sh[i] = A.pre[((size_t)n * A.Ci + ci) * 2 + 1]; }
        }
    };
    auto store = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * W2_PT;
        const bool pvv = pt + c4 < P;                   // P % 4 == 0: all four or none
#pragma unroll
        for (int i = 0; i < W2_ND; ++i) {
            const bool ok = pvv && (co0 + drow[i] < A.Co);
            float4 v;
            v.x = ok ? fmaf(k0[i], rg[i].x, fmaf(k1[i], ra[i].x, k2[i])) : 0.f;
            v.y = ok ? fmaf(k0[i], rg[i].y, fmaf(k1[i], ra[i].y, k2[i])) : 0.f;
            v.z = ok ? fmaf(k0[i], rg[i].z, fmaf(k1[i], ra[i].z, k2[i])) : 0.f;
            v.w = ok ? fmaf(k0[i], rg[i].w, fmaf(k1[i], ra[i].w, k2[i])) : 0.f;
            *reinterpret_cast<float4*>(&Dl[drow[i] * W2_LD + c4]) = v;
        }
#pragma unroll
        for (int i = 0; i < W2_NX; ++i) {
            const bool ok = pvv && (ci0 + xrow_[i] < A.Ci);
            float4 v = rx[i];
            if (A.pre != nullptr) {
                v.x = act_fwd(fmaf(sc[i], v.x, sh[i]), A.pre_act);
                v.y = act_fwd(fmaf(sc[i], v.y, sh[i]), A.pre_act);
                v.z = act_fwd(fmaf(sc[i], v.z, sh[i]), A.pre_act);
                v.w = act_fwd(fmaf(sc[i], v.w, sh[i]), A.pre_act);
            }
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            *reinterpret_cast<float4*>(&Xl[xrow_[i] * W2_LD + c4]) = v;
        }
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&]() {
#pragma unroll
        for (int s4 = 0; s4 < W2_PT / 16; ++s4) {
            float4 av[2], bv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                av[i] = *reinterpret_cast<const float4*>(&Dl[((2 * wave + i) * 16 + r) * W2_LD + s4 * 16 + 4 * q]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bv[j] = *reinterpret_cast<const float4*>(&Xl[(j * 16 + r) * W2_LD + s4 * 16 + 4 * q]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    int c = blockIdx.x;
    if (c < total) {
        fetch(c);
        store(c);
        __syncthreads();
        for (; c < total; c += A.groups) {
            const int cn = c + A.groups;
            if (cn < total) fetch(cn);
            compute();
            __syncthreads();                 // everyone done reading this chunk
            if (cn < total) { store(cn); __syncthreads(); }
        }
    }

    // D[i = co][j = ci]: lane (q, r), reg e -> co = tile*16 + 4q + e, ci = tile*16 + r
    float* out = A.wpartial + (size_t)blockIdx.x * A.Co * A.Ci;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int oc = co0 + (2 * wave + i) * 16 + 4 * q + e, ic = ci0 + j * 16 + r;
                if (oc < A.Co && ic < A.Ci) out[(size_t)oc * A.Ci + ic] = acc[i][j][e];
            }
}

// ---------------------------------------------------------------------------------------
// Split-bf16 form of the tiled weight-gradient kernel (the default).
// The fp32 MFMA runs at 1/16 of the bf16 rate on gfx950 and the tiled kernel above spends most of
// each chunk inside its 128 fp32 MFMAs per wave.  Here both operands are split while they are staged
// into LDS, v = hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant bits), and every
// 16x16x32 tile product is three bf16 MFMAs, hi*hi + hi*lo + lo*hi, accumulated in fp32: 48 MFMAs of
// 16 cycles per chunk instead of 128 of 32.  The dropped lo*lo term and the truncation of lo are
// ~2^-16 relative per product -- about 1e-5 on dW even under heavy cancellation, two orders below the
// parity tolerance -- and dW feeds only the optimizer, so the error is not amplified by later
// layers (the same split on the forward / data-gradient GEMMs moves the logits by 2.6e-3:
// tests/exp_split_precision.py; those stay fp32).
// LDS image per operand plane: [row][64 voxels] bf16, 144-byte pitch (conflict-free b128 reads).
// ---------------------------------------------------------------------------------------
constexpr int W3_LD = 72;          // bf16 elements per LDS row

// v = hi + mid + lo: three bf16 terms = all 24 significant bits of an fp32 value (mid = bf16(v - hi), lo = bf16(v - hi - mid));
// the two-term kernels (NS = 2, option bwd_terms = 2) keep hi and mid
__device__ __forceinline__ void split_bf16x4(const float (&v)[4], bf16x4& hi, bf16x4& mid, bf16x4& lo) {
    uint2 H, M, L;                                  // pairs (v0, v1), (v2, v3): x3d_split3_pair, common.h
    x3d_split3_pair(v[0], v[1], H.x, M.x, L.x);
    x3d_split3_pair(v[2], v[3], H.y, M.y, L.y);
    hi = __builtin_bit_cast(bf16x4, H);
    mid = __builtin_bit_cast(bf16x4, M);
    lo = __builtin_bit_cast(bf16x4, L);
}

// CO x CI = output tile of a workgroup (dY channels x input channels): 128 x 64 for the wide layers, 64-row and
// 32-column variants for the narrow ones (stages 1-2) so that padding rows are not staged for nothing.
// GATHER: the input of a stride-(1,2,2) convolution (the downsample branch): its four voxels per slot are read at
// (t, 2 ho, 2 wo) of the full-resolution input instead of as one float4.
// NS: bf16 terms per fp32 operand: 3 = six MFMA products, fp32-level (default); 2 = three products, ~2^-16 per product.
template <int CO, int CI, bool GATHER, bool MX, int NS>
__device__ __forceinline__ void wgrad3_body(const WgArgs& A, const int grp, const int blk) {
    const int ga_bf = MX ? A.ga_bf : 0, x_bf = (MX && !GATHER) ? A.x_bf : 0;
    constexpr int ND = CO / 16, NX = CI / 16;            // staged float4 slots per thread (dY, input)
    constexpr int MW = CO / 64, NW = CI / 16;            // 16x16 tiles per wave: MW (dY) x NW (input)
    __shared__ __attribute__((aligned(16))) __bf16 Dh[CO * W3_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Dm[CO * W3_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Dlo[(NS == 3 ? CO : 1) * W3_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xh[CI * W3_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xm[CI * W3_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xlo[(NS == 3 ? CI : 1) * W3_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int co0 = (blk / A.cib) * CO, ci0 = (blk % A.cib) * CI;
    const int P = A.P;
    const int cps = (P + W2_PT - 1) / W2_PT;            // chunks per sample
    const int total = A.N * cps;

    const int c4 = (tid & 15) * 4;
    const int row0 = tid >> 4;                          // staging rows row0 + 16 i

    float4 rg[ND], ra[ND], rx[NX];
    float k0[ND], k1[ND], k2[ND], sc[NX], sh[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { sc[i] = 1.f; sh[i] = 0.f; }

    auto fetch = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * W2_PT;
        const int pc = min(pt + c4, P - 4);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int co = min(co0 + row0 + 16 * i, A.Co - 1);
            const size_t base = ((size_t)n * A.Co + co) * (size_t)P + pc;
            rg[i] = ldx4_raw(A.g, base, ga_bf);
            ra[i] = ldx4_raw(A.a, base, ga_bf);
            const float* cb = A.cb + ((size_t)n * A.Co + co) * 3;
            k0[i] = cb[0]; k1[i] = cb[1]; k2[i] = cb[2];
        }
        int goff[4] = {0, 0, 0, 0};
        if (GATHER) {                                    // input offsets of this slot's four output voxels (row independent)
            const int hw = A.Ho * A.Wo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = pc + e;
                const int t = p / hw, rem = p - t * hw;
                const int ho = rem / A.Wo, wo = rem - ho * A.Wo;
                goff[e] = (t * A.H + 2 * ho) * A.W + 2 * wo;
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int ci = min(ci0 + row0 + 16 * i, A.Ci - 1);
            const float* px = A.x + ((size_t)n * A.Ci + ci) * (size_t)A.Pin;
            if (GATHER) rx[i] = make_float4(px[goff[0]], px[goff[1]], px[goff[2]], px[goff[3]]);
            else rx[i] = ldx4_raw(A.x, ((size_t)n * A.Ci + ci) * (size_t)A.Pin + pc, x_bf);
            if (A.pre != nullptr) {
                const float2 p2 = *reinterpret_cast<const float2*>(A.pre + ((size_t)n * A.Ci + ci) * 2);
                sc[i] = p2.x; sh[i] = p2.y;
            }
        }
    };
    auto store = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * W2_PT;
        const bool pvv = pt + c4 < P;                   // P % 4 == 0: all four or none
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = pvv && (co0 + row0 + 16 * i < A.Co);
            const float4 gw = widen4(rg[i], ga_bf), aw = widen4(ra[i], ga_bf);
            const float gv[4] = {gw.x, gw.y, gw.z, gw.w}, av[4] = {aw.x, aw.y, aw.z, aw.w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? fmaf(k0[i], gv[e], fmaf(k1[i], av[e], k2[i])) : 0.f;
            bf16x4 hi, mid, lo;
            split_bf16x4(v, hi, mid, lo);
            *reinterpret_cast<bf16x4*>(&Dh[(row0 + 16 * i) * W3_LD + c4]) = hi;
            *reinterpret_cast<bf16x4*>(&Dm[(row0 + 16 * i) * W3_LD + c4]) = mid;
            if (NS == 3) *reinterpret_cast<bf16x4*>(&Dlo[(row0 + 16 * i) * W3_LD + c4]) = lo;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool ok = pvv && (ci0 + row0 + 16 * i < A.Ci);
            const float4 xw = widen4(rx[i], x_bf);
            float v[4] = {xw.x, xw.y, xw.z, xw.w};
            if (A.pre != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(fmaf(sc[i], v[e], sh[i]), A.pre_act);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            bf16x4 hi, mid, lo;
            split_bf16x4(v, hi, mid, lo);
            *reinterpret_cast<bf16x4*>(&Xh[(row0 + 16 * i) * W3_LD + c4]) = hi;
            *reinterpret_cast<bf16x4*>(&Xm[(row0 + 16 * i) * W3_LD + c4]) = mid;
            if (NS == 3) *reinterpret_cast<bf16x4*>(&Xlo[(row0 + 16 * i) * W3_LD + c4]) = lo;
        }
    };

    f32x4 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // 16x16x32 bf16 fragments: lane (r, q) holds k = 8q .. 8q+7 of row r (A: dY channel, B: input channel)
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < W2_PT / 32; ++s) {
            bf16x8 ah[MW], am[MW], al[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                const int off = ((MW * wave + i) * 16 + r) * W3_LD + s * 32 + 8 * q;
                ah[i] = *reinterpret_cast<const bf16x8*>(&Dh[off]);
                am[i] = *reinterpret_cast<const bf16x8*>(&Dm[off]);
                if (NS == 3) al[i] = *reinterpret_cast<const bf16x8*>(&Dlo[off]);
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int off = (j * 16 + r) * W3_LD + s * 32 + 8 * q;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Xh[off]);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&Xm[off]);
                bf16x8 bl;
                if (NS == 3) bl = *reinterpret_cast<const bf16x8*>(&Xlo[off]);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (NS == 3) {                     // smallest terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm, acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    int c = grp;
    if (c < total) {
        fetch(c);
        store(c);
        __syncthreads();
        for (; c < total; c += A.groups) {
            const int cn = c + A.groups;
            if (cn < total) fetch(cn);
            compute();
            __syncthreads();                 // everyone done reading this chunk
            if (cn < total) { store(cn); __syncthreads(); }
        }
    }

    // D[i = co][j = ci]: lane (q, r), reg e -> co = tile*16 + 4q + e, ci = tile*16 + r
    float* out = A.wpartial + (size_t)grp * A.Co * A.Ci;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int oc = co0 + (MW * wave + i) * 16 + 4 * q + e, ic = ci0 + j * 16 + r;
                if (oc < A.Co && ic < A.Ci) out[(size_t)oc * A.Ci + ic] = acc[i][j][e];
            }
}

// Wide-tile form for the large-channel layers (stages 3-4, conv5; dense input): 8 waves, a (256 x 96) or (128 x 224) block
// of dW per workgroup, so that a stage-3 weight (216 x 96 / 96 x 216) is ONE block and every voxel chunk is staged once
// instead of once per 128 x 64 block (4 blocks at stage 3, 12 at stage 4: the batched launches fetched 2.2x their
// algorithmic bytes, profiles/r02/a_traffic.json, and split the same values four times).  Waves form a 4 (dY) x 2 (input)
// grid: MW x NW sixteen-row tiles per wave keep the LDS fragment reads per MFMA at the 128 x 64 kernel's level.
template <int CO, int CI, bool MX, int NS>
__device__ __forceinline__ void wgrad4_body(const WgArgs& A, const int grp, const int blk) {
    const int ga_bf = MX ? A.ga_bf : 0, x_bf = MX ? A.x_bf : 0;
    constexpr int NT = 512, RP = NT / 16;                // 32 rows staged per pass
    constexpr int ND = CO / RP, NX = CI / RP;
    constexpr int WCO = 4, WCI = 2;
    constexpr int MW = CO / 16 / WCO, NW = CI / 16 / WCI;
    static_assert(CO % (16 * WCO) == 0 && CI % (16 * WCI) == 0 && CO % RP == 0 && CI % RP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) __bf16 wl4[];
    __bf16* Dh = wl4;                                     // NS planes of dY, then NS planes of the input
    __bf16* Dm = Dh + CO * W3_LD;
    __bf16* Dlo = Dh + (NS - 1) * CO * W3_LD;
    __bf16* Xh = Dh + NS * CO * W3_LD;
    __bf16* Xm = Xh + CI * W3_LD;
    __bf16* Xlo = Xh + (NS - 1) * CI * W3_LD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int co0 = (blk / A.cib) * CO, ci0 = (blk % A.cib) * CI;
    const int P = A.P;
    const int cps = (P + W2_PT - 1) / W2_PT;
    const int total = A.N * cps;
    const int c4 = (tid & 15) * 4;
    const int row0 = tid >> 4;

    float4 rg[ND], ra[ND], rx[NX];
    float k0[ND], k1[ND], k2[ND], sc[NX], sh[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { sc[i] = 1.f; sh[i] = 0.f; }

    auto fetch = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * W2_PT;
        const int pc = min(pt + c4, P - 4);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int co = min(co0 + row0 + RP * i, A.Co - 1);
            const size_t base = ((size_t)n * A.Co + co) * (size_t)P + pc;
            rg[i] = ldx4_raw(A.g, base, ga_bf);
            ra[i] = ldx4_raw(A.a, base, ga_bf);
            const float* cb = A.cb + ((size_t)n * A.Co + co) * 3;
            k0[i] = cb[0]; k1[i] = cb[1]; k2[i] = cb[2];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int ci = min(ci0 + row0 + RP * i, A.Ci - 1);
            rx[i] = ldx4_raw(A.x, ((size_t)n * A.Ci + ci) * (size_t)A.Pin + pc, x_bf);
            if (A.pre != nullptr) {
                const float2 p2 = *reinterpret_cast<const float2*>(A.pre + ((size_t)n * A.Ci + ci) * 2);
                sc[i] = p2.x; sh[i] = p2.y;
            }
        }
    };
    auto store = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * W2_PT;
        (void)n;
        const bool pvv = pt + c4 < P;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = pvv && (co0 + row0 + RP * i < A.Co);
            const float4 gw = widen4(rg[i], ga_bf), aw = widen4(ra[i], ga_bf);
            const float gv[4] = {gw.x, gw.y, gw.z, gw.w}, av[4] = {aw.x, aw.y, aw.z, aw.w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? fmaf(k0[i], gv[e], fmaf(k1[i], av[e], k2[i])) : 0.f;
            bf16x4 hi, mid, lo;
            split_bf16x4(v, hi, mid, lo);
            *reinterpret_cast<bf16x4*>(&Dh[(row0 + RP * i) * W3_LD + c4]) = hi;
            *reinterpret_cast<bf16x4*>(&Dm[(row0 + RP * i) * W3_LD + c4]) = mid;
            if (NS == 3) *reinterpret_cast<bf16x4*>(&Dlo[(row0 + RP * i) * W3_LD + c4]) = lo;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool ok = pvv && (ci0 + row0 + RP * i < A.Ci);
            const float4 xw = widen4(rx[i], x_bf);
            float v[4] = {xw.x, xw.y, xw.z, xw.w};
            if (A.pre != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(fmaf(sc[i], v[e], sh[i]), A.pre_act);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            bf16x4 hi, mid, lo;
            split_bf16x4(v, hi, mid, lo);
            *reinterpret_cast<bf16x4*>(&Xh[(row0 + RP * i) * W3_LD + c4]) = hi;
            *reinterpret_cast<bf16x4*>(&Xm[(row0 + RP * i) * W3_LD + c4]) = mid;
            if (NS == 3) *reinterpret_cast<bf16x4*>(&Xlo[(row0 + RP * i) * W3_LD + c4]) = lo;
        }
    };

    f32x4 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wco0 = MW * (wave % WCO), wci0 = NW * (wave / WCO);

    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < W2_PT / 32; ++s) {
            bf16x8 ah[MW], am[MW], al[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                const int off = ((wco0 + i) * 16 + r) * W3_LD + s * 32 + 8 * q;
                ah[i] = *reinterpret_cast<const bf16x8*>(&Dh[off]);
                am[i] = *reinterpret_cast<const bf16x8*>(&Dm[off]);
                if (NS == 3) al[i] = *reinterpret_cast<const bf16x8*>(&Dlo[off]);
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int off = ((wci0 + j) * 16 + r) * W3_LD + s * 32 + 8 * q;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Xh[off]);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&Xm[off]);
                bf16x8 bl;
                if (NS == 3) bl = *reinterpret_cast<const bf16x8*>(&Xlo[off]);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (NS == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm, acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    int c = grp;
    if (c < total) {
        fetch(c);
        store(c);
        __syncthreads();
        for (; c < total; c += A.groups) {
            const int cn = c + A.groups;
            if (cn < total) fetch(cn);
            compute();
            __syncthreads();
            if (cn < total) { store(cn); __syncthreads(); }
        }
    }
    float* out = A.wpartial + (size_t)grp * A.Co * A.Ci;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int oc = co0 + (wco0 + i) * 16 + 4 * q + e, ic = ci0 + (wci0 + j) * 16 + r;
                if (oc < A.Co && ic < A.Ci) out[(size_t)oc * A.Ci + ic] = acc[i][j][e];
            }
}

struct WgBatch;
template <int CO, int CI, bool MX, int NS>
__global__ __launch_bounds__(512, 2) void pw_wgrad4_batch_kernel(const WgBatch B);

template <int CO, int CI, bool GATHER, bool MX, int NS>
__global__ __launch_bounds__(256, 2) void pw_wgrad3_kernel(const WgArgs A) {
    wgrad3_body<CO, CI, GATHER, MX, NS>(A, blockIdx.x, blockIdx.y);
}

// Every weight gradient of a backward pass that uses the same tile variant in one launch (up to WB_MAX jobs, passed by
// value): nothing downstream of a weight gradient runs before the optimizer, so the host postpones them; one launch
// keeps all CUs busy across the short per-layer problems (512 workgroups x 3 voxel chunks at stage 3) instead of
// paying a ramp and a tail per layer.  Workgroup -> (job, group, channel block): job-local ids keep the
// group-major order of the single launch (groups % 8 == 0, so a group's channel blocks still share one XCD's L2).
constexpr int WB_MAX = 24;
struct WgBatch {
    WgArgs job[WB_MAX];
    int wg0[WB_MAX + 1];
    int njobs;
};

template <int CO, int CI, bool MX, int NS>
__global__ __launch_bounds__(512, 2) void pw_wgrad4_batch_kernel(const WgBatch B) {
    int j = 0;
    while (j + 1 < B.njobs && (int)blockIdx.x >= B.wg0[j + 1]) ++j;
    const int local = (int)blockIdx.x - B.wg0[j];
    const int groups = B.job[j].groups;
    wgrad4_body<CO, CI, MX, NS>(B.job[j], local % groups, local / groups);
}

template <int CO, int CI, bool GATHER, bool MX, int NS>
__global__ __launch_bounds__(256, 2) void pw_wgrad3_batch_kernel(const WgBatch B) {
    int j = 0;
    while (j + 1 < B.njobs && (int)blockIdx.x >= B.wg0[j + 1]) ++j;
    const int local = (int)blockIdx.x - B.wg0[j];
    const int groups = B.job[j].groups;
    wgrad3_body<CO, CI, GATHER, MX, NS>(B.job[j], local % groups, local / groups);
}

static bool wgrad2_ok(int P, long long Pin, int Co, int Ci, bool strided) {
    if (strided) return (P % 4 == 0) && !x3d_opt(X3D_OPT_WGRAD_F32);           // gathered input: split-bf16 kernel only
    return (P % 4 == 0) && (Pin % 4 == 0);
}
// workgroup tile of the split-bf16 kernel for a Co x Ci weight
static int wg3_co(int Co) { return Co > 64 ? 128 : 64; }
static int wg3_ci(int Ci) { return Ci > 32 ? 64 : 32; }
// wide tiles (wgrad4_body, batched launches only): 1 = 256 x 96 (many dY channels), 2 = 128 x 224 (many input channels)
static int wg4_kind(int Co, int Ci, bool dense) {
    const bool off = x3d_opt(X3D_OPT_NO_WGRAD4) != 0;
    if (off || !dense) return 0;
    if (Co > 128 && Ci > 64) return Co >= Ci ? 1 : 2;
    if (Co > 64 && Ci > 128) return 2;
    return 0;
}

// tiled = 1: pw_wgrad2_kernel (cob x cib blocks of 128 x 64), else the direct-load kernel
static void wgrad_plan(int N, int P, int Co, int Ci, bool dense, int* tiled, int* groups, int* cob, int* cib,
                       int* ct_run, int* it_run) {
    if (wgrad2_ok(P, dense ? P : 1, Co, Ci, !dense)) {
        *tiled = 1;
        const bool f32 = x3d_opt(X3D_OPT_WGRAD_F32) != 0;        // exact fp32-MFMA kernel: fixed 128 x 64 tiles
        *cob = cdiv(Co, f32 ? W2_CO : wg3_co(Co)); *cib = cdiv(Ci, f32 ? W2_CI : wg3_ci(Ci));
        *ct_run = 8; *it_run = 4;
        const int chunks = N * cdiv(P, W2_PT);
        // ~8 chunks per workgroup, <= ~256 workgroups per conv: the default path launches all weight gradients of a
        // backward pass together (x3d_pw_bwd_weight_batch), so the chip is filled by the batch and fewer, longer
        // workgroups mean fewer partials to write and sum (measured 3/640 -> 8/256: +1 % on the step; for a conv
        // launched on its own 3/640 is the better choice)
        const int cpw = x3d_opt(X3D_OPT_WG_CPW);
        const int wcap = x3d_opt(X3D_OPT_WG_CAP);
        int g = cdiv(chunks, cpw);
        const int cap = wcap / ((*cob) * (*cib)) > 16 ? wcap / ((*cob) * (*cib)) : 16;
        if (g > cap) g = cap;
        if (g < 1) g = 1;
        // XCD-aware: workgroup id = group + groups * (co/ci block) and workgroups go round-robin over the 8 XCDs, so
        // with groups % 8 == 0 every (co, ci) block of one voxel chunk runs on the same XCD and the repeated reads
        // of that chunk's dY / x rows hit its L2 (measured before: 2-4x the algorithmic bytes fetched from HBM)
        if (g >= 8) g &= ~7;
        *groups = g;
        return;
    }
    *tiled = 0;
    const int cot = cdiv(Co, 16), cit = cdiv(Ci, 16);
    *cob = cdiv(cot, 4); *cib = cdiv(cit, 4);
    *ct_run = cdiv(cot, *cob); *it_run = cdiv(cit, *cib);
    const int units = N * cdiv(P, WG_UNIT);
    int g = cdiv(units, 8);                       // >= 2 units per wave when there is enough work
    const int cap = 1024 / ((*cob) * (*cib)) > 64 ? 1024 / ((*cob) * (*cib)) : 64;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    if (g >= 8) g &= ~7;                          // same XCD for all channel blocks of a voxel group (see above)
    *groups = g;
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial,
                                                              float* __restrict__ out, int groups, int n) {
    // 64 outputs per block; the 4 waves take interleaved quarters of the groups (each load is a
    // 256-B coalesced row segment); fixed summation order -> bitwise reproducible
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < n) {
        int g = wave;
        for (; g + 60 < groups; g += 64) {           // round 4: sixteen rows in flight per wave (512 groups: 8 round trips)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(g + 4 * u) * n + i];
#pragma unroll
            for (int u = 0; u < 16; u += 8) {          // same grouping of the additions as the eight-row loop below
                s += ((double)v[u] + (double)v[u + 1]) + ((double)v[u + 2] + (double)v[u + 3]);
                s += ((double)v[u + 4] + (double)v[u + 5]) + ((double)v[u + 6] + (double)v[u + 7]);
            }
        }
        for (; g + 28 < groups; g += 32) {           // eight rows in flight per wave (512 groups: 16 round trips instead of 32)
            const float v0 = partial[(size_t)g * n + i], v1 = partial[(size_t)(g + 4) * n + i];
            const float v2 = partial[(size_t)(g + 8) * n + i], v3 = partial[(size_t)(g + 12) * n + i];
            const float v4 = partial[(size_t)(g + 16) * n + i], v5 = partial[(size_t)(g + 20) * n + i];
            const float v6 = partial[(size_t)(g + 24) * n + i], v7 = partial[(size_t)(g + 28) * n + i];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
            s += ((double)v4 + (double)v5) + ((double)v6 + (double)v7);
        }
        for (; g + 12 < groups; g += 16) {
            const float v0 = partial[(size_t)g * n + i], v1 = partial[(size_t)(g + 4) * n + i];
            const float v2 = partial[(size_t)(g + 8) * n + i], v3 = partial[(size_t)(g + 12) * n + i];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; g < groups; g += 4) s += (double)partial[(size_t)g * n + i];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < n) out[i] = (float)((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// All group reductions of a backward pass in one launch: up to RB_MAX jobs per launch, passed by value (kernel
// arguments); a workgroup finds its job by a scan of the workgroup prefix table.  Same per-output summation order as
// reduce_partials_kernel -> bitwise identical results.
constexpr int RB_MAX = 96;
struct ReduceBatch {
    const float* partial[RB_MAX];
    float* out[RB_MAX];
    int groups[RB_MAX];
    int n[RB_MAX];
    int wg0[RB_MAX + 1];
    int njobs;
};

__global__ __launch_bounds__(256) void reduce_partials_batch_kernel(const ReduceBatch B) {
    __shared__ double red[4][64];
    int j = 0;
    while (j + 1 < B.njobs && (int)blockIdx.x >= B.wg0[j + 1]) ++j;          // uniform: scalar loads
    const float* __restrict__ partial = B.partial[j];
    float* __restrict__ out = B.out[j];
    const int groups = B.groups[j], n = B.n[j];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = ((int)blockIdx.x - B.wg0[j]) * 64 + lane;
    double s = 0.0;
    if (i < n) {
        int g = wave;
        for (; g + 60 < groups; g += 64) {           // round 4: sixteen rows in flight per wave (512 groups: 8 round trips)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(g + 4 * u) * n + i];
#pragma unroll
            for (int u = 0; u < 16; u += 8) {          // same grouping of the additions as the eight-row loop below
                s += ((double)v[u] + (double)v[u + 1]) + ((double)v[u + 2] + (double)v[u + 3]);
                s += ((double)v[u + 4] + (double)v[u + 5]) + ((double)v[u + 6] + (double)v[u + 7]);
            }
        }
        for (; g + 28 < groups; g += 32) {           // eight rows in flight per wave (512 groups: 16 round trips instead of 32)
            const float v0 = partial[(size_t)g * n + i], v1 = partial[(size_t)(g + 4) * n + i];
            const float v2 = partial[(size_t)(g + 8) * n + i], v3 = partial[(size_t)(g + 12) * n + i];
            const float v4 = partial[(size_t)(g + 16) * n + i], v5 = partial[(size_t)(g + 20) * n + i];
            const float v6 = partial[(size_t)(g + 24) * n + i], v7 = partial[(size_t)(g + 28) * n + i];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
            s += ((double)v4 + (double)v5) + ((double)v6 + (double)v7);
        }
        for (; g + 12 < groups; g += 16) {
            const float v0 = partial[(size_t)g * n + i], v1 = partial[(size_t)(g + 4) * n + i];
            const float v2 = partial[(size_t)(g + 8) * n + i], v3 = partial[(size_t)(g + 12) * n + i];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; g < groups; g += 4) s += (double)partial[(size_t)g * n + i];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < n) out[i] = (float)((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

}  // namespace

extern "C" int x3d_reduce_partials_batch(const float* const* partials, float* const* outs, const int* groups,
                                         const int* ns, int njobs, void* stream) {
    X3D_CHECK_ARG(partials && outs && groups && ns && njobs > 0);
    for (int j0 = 0; j0 < njobs; j0 += RB_MAX) {
        ReduceBatch B;
        B.njobs = min(RB_MAX, njobs - j0);
        int wg = 0;
        for (int j = 0; j < B.njobs; ++j) {
            X3D_CHECK_ARG(partials[j0 + j] && outs[j0 + j] && groups[j0 + j] > 0 && ns[j0 + j] > 0);
            B.partial[j] = partials[j0 + j]; B.out[j] = outs[j0 + j]; B.groups[j] = groups[j0 + j]; B.n[j] = ns[j0 + j];
            B.wg0[j] = wg;
            wg += cdiv(ns[j0 + j], 64);
        }
        B.wg0[B.njobs] = wg;
        hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, B);
        X3D_LAUNCH_CHECK();
    }
    return X3D_OK;
}

extern "C" int x3d_pw_tiles(int N, int K, int M, int P, int dense) {
    int variant, tiles, mb, mt;
    pw_plan(N, K, M, P, dense != 0, &variant, &tiles, &mb, &mt);
    return tiles;
}

// tiles of the forward's `partial`: the large-channel forward kernel (packed weights, dense) works on 32-voxel items
extern "C" int x3d_pw_fwd_tiles(int N, int Cin, int Cout, int P, int dense, int packed) {
    if (packed && dense && x3d_pw6_ok(Cin, Cout, P)) return x3d_pw6_tiles(P);
    if (packed && dense && x3d_pwfs_ok(Cin, Cout, P)) return x3d_pwfs_tiles(N, P);
    return x3d_pw_tiles(N, Cin, Cout, P, dense);
}

// tiles of the data gradient's `partial` (Cout = its K, Cin = its M)
extern "C" int x3d_pw_bwd_tiles(int N, int Cin, int Cout, int P, int packed, int mx) {
    if (packed && x3d_pw7_ok(Cout, Cin, P, mx)) return x3d_pw6_tiles(P);
    return x3d_pw_tiles(N, Cout, Cin, P, 1);
}

extern "C" int x3d_pw_wants_packed(int K, int M) { (void)K; (void)M; return 1; }

// both orientations carry three bf16 planes (hi, mid, lo) behind the fp32 image since ABI 6 (three-term backward GEMMs)
extern "C" size_t x3d_pw_pack_floats(int K, int M, int transposed) { (void)transposed; return pack_floats(K, M, 3); }
extern "C" size_t x3d_pw_pack_items(int K, int M, int transposed) { (void)transposed; return pack_items(K, M, 3); }

extern "C" int x3d_pw_pack(const float* w, float* wp, int Cout, int Cin, int transposed, void* stream) {
    X3D_CHECK_ARG(w && wp && Cout > 0 && Cin > 0);
    // forward:    A[row = co][k = ci] = w[co*Cin + ci]   (M = Cout, K = Cin)
    // transposed: A[row = ci][k = co] = w[co*Cin + ci]   (M = Cin,  K = Cout)   (backward-data; + split-bf16 planes)
    const int M = transposed ? Cin : Cout, K = transposed ? Cout : Cin;
    const int ldm = transposed ? 1 : Cin, ldk = transposed ? Cin : 1;
    const int mtiles = cdiv(M, 16), kgroups = cdiv(K, 16);
    hipLaunchKernelGGL(pw_pack_kernel, dim3((unsigned)cdiv((int)pack_items(K, M, 3), 256)), dim3(256), 0,
                       (hipStream_t)stream, w, wp, M, K, ldm, ldk, mtiles, kgroups, 3);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" size_t x3d_pw_pack_job_bytes(void) { return sizeof(PackJob); }

extern "C" int x3d_pw_pack_batch(const void* jobs, const int* wg_job, int n_workgroups, void* stream) {
    X3D_CHECK_ARG(jobs && wg_job && n_workgroups > 0);
    hipLaunchKernelGGL(pw_pack_batch_kernel, dim3(n_workgroups), dim3(256), 0, (hipStream_t)stream,
                       (const PackJob*)jobs, wg_job);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_pw_fwd(const void* x, const float* w, const float* wpacked, void* y, int N, int Cin, int Cout,
                          int T, int H, int W, int strideHW, const float* pre, int pre_act, float* partial, int mx,
                          void* stream) {
    X3D_CHECK_ARG(x && w && y);
    X3D_CHECK_ARG((mx & ~(X3D_MX_X | X3D_MX_Y)) == 0);
    const int x_bf = (mx & X3D_MX_X) != 0, y_bf = (mx & X3D_MX_Y) != 0;
    X3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    X3D_CHECK_ARG(N <= 65535);
    PwArgs A = {};
    const int Ho = strideHW == 2 ? (H - 1) / 2 + 1 : H, Wo = strideHW == 2 ? (W - 1) / 2 + 1 : W;
    A.x = (const float*)x; A.a = nullptr; A.cin = pre; A.w = w; A.wp = wpacked; A.w_ldk = 1; A.w_ldm = Cin; A.y = (float*)y;
    A.x_bf = x_bf; A.y_bf = y_bf;
    A.N = N; A.K = Cin; A.M = Cout; A.P = T * Ho * Wo; A.Pin = (long long)T * H * W;
    A.in_act = pre_act; A.strided = strideHW == 2; A.T = T; A.H = H; A.W = W; A.Ho = Ho; A.Wo = Wo;
    A.partial = partial; A.addend = nullptr; A.addend_stride = 1;
    hipStream_t s = (hipStream_t)stream;
    if (wpacked != nullptr && strideHW == 1 && x3d_pw6_ok(Cin, Cout, A.P))
        return x3d_pw6_launch(x, pre, wpacked, y, partial, N, Cin, Cout, A.P, pre_act, x_bf, y_bf, s);
    if (wpacked != nullptr && strideHW == 1 && x3d_pwfs_ok(Cin, Cout, A.P))
        return x3d_pwfs_launch(x, pre, wpacked, y, partial, N, Cin, Cout, A.P, pre_act, x_bf, y_bf, s);
    if (pre) return launch_pw<IN_AFFACT, EPI_STATS>(A, s);
    return launch_pw<IN_RAW, EPI_STATS>(A, s);
}

extern "C" int x3d_pw_bwd_data(const void* g, const void* a, const float* cb, const float* w,
                               const float* wpacked_t, void* out,
                               int N, int Cin, int Cout, int T, int H, int W, const void* x,
                               const float* pre, int pre_act, const float* addend, int addend_stride,
                               float* partial, int mx, void* stream) {
    X3D_CHECK_ARG(g && a && cb && w && out);
    X3D_CHECK_ARG((mx & ~(X3D_MX_X | X3D_MX_Y | X3D_MX_GA)) == 0);
    const int ga_bf = (mx & X3D_MX_GA) != 0, y_bf = (mx & X3D_MX_Y) != 0, ex_bf = (mx & X3D_MX_X) != 0 && pre != nullptr;
    X3D_CHECK_ARG(N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(addend_stride == 1 || addend_stride == 2);
    X3D_CHECK_ARG((pre == nullptr) || (x != nullptr));
    PwArgs A = {};
    A.x = (const float*)g; A.a = (const float*)a; A.cin = cb; A.w = w; A.wp = wpacked_t; A.w_ldk = Cin; A.w_ldm = 1;
    A.y = (float*)out;
    A.N = N; A.K = Cout; A.M = Cin; A.P = T * H * W; A.Pin = A.P;
    A.strided = 0; A.T = T; A.H = H; A.W = W;
    A.Ho = (H - 1) / 2 + 1; A.Wo = (W - 1) / 2 + 1;
    A.partial = partial; A.ex = (const float*)x; A.ecoef = pre; A.e_act = pre_act;
    A.x_bf = ga_bf; A.y_bf = y_bf; A.ex_bf = ex_bf;
    A.addend = addend; A.addend_stride = addend_stride;
    hipStream_t s = (hipStream_t)stream;
    if (wpacked_t != nullptr && x3d_pw7_ok(Cout, Cin, A.P, mx))
        return x3d_pw7_launch(g, a, cb, wpacked_t, out, partial, pre ? 1 : 0, x, nullptr, pre, pre_act, addend, addend_stride,
                              N, Cout, Cin, T, H, W, ga_bf, y_bf, ex_bf, s);
    if (pre) return launch_pw<IN_BNBWD, EPI_ACTBWD>(A, s);
    return launch_pw<IN_BNBWD, EPI_PLAIN>(A, s);
}

extern "C" int x3d_pw_bwd_data_res(const void* g, const void* a, const float* cb, const float* w,
                                   const float* wpacked_t, float* out, int N, int Cin, int Cout, int T, int H, int W,
                                   const float* res_out, const float* res_raw, const float* addend, int addend_stride,
                                   float* partial, int mx, void* stream) {
    X3D_CHECK_ARG(g && a && cb && w && out && res_out && res_raw && partial);
    X3D_CHECK_ARG((mx & ~X3D_MX_GA) == 0);
    const int ga_bf = (mx & X3D_MX_GA) != 0;
    X3D_CHECK_ARG(N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(addend_stride == 1 || addend_stride == 2);
    PwArgs A = {};
    A.x = (const float*)g; A.a = (const float*)a; A.cin = cb; A.w = w; A.wp = wpacked_t; A.w_ldk = Cin; A.w_ldm = 1; A.y = out;
    A.x_bf = ga_bf;
    A.N = N; A.K = Cout; A.M = Cin; A.P = T * H * W; A.Pin = A.P;
    A.strided = 0; A.T = T; A.H = H; A.W = W;
    A.Ho = (H - 1) / 2 + 1; A.Wo = (W - 1) / 2 + 1;
    A.partial = partial; A.ex = res_raw; A.emask = res_out; A.ecoef = nullptr; A.e_act = X3D_ACT_RELU;
    A.addend = addend; A.addend_stride = addend_stride;
    if (wpacked_t != nullptr && x3d_pw7_ok(Cout, Cin, A.P, mx))
        return x3d_pw7_launch(g, a, cb, wpacked_t, out, partial, 2, res_raw, res_out, nullptr, X3D_ACT_RELU, addend,
                              addend_stride, N, Cout, Cin, T, H, W, ga_bf, 0, 0, (hipStream_t)stream);
    return launch_pw<IN_BNBWD, EPI_RESBWD>(A, (hipStream_t)stream);
}

extern "C" int x3d_pw_wgrad_groups(int N, int P, int Cout, int Cin, int strideHW) {
    int tiled, g, cob, cib, ct, it;
    wgrad_plan(N, P, Cout, Cin, strideHW == 1, &tiled, &g, &cob, &cib, &ct, &it);
    return g;
}

extern "C" int x3d_pw_bwd_weight(const void* g, const void* a, const float* cb, const void* x,
                                 const float* pre, int pre_act, float* wpartial, int N, int Cin, int Cout,
                                 int T, int H, int W, int strideHW, int mx, void* stream) {
    X3D_CHECK_ARG(g && a && cb && x && wpartial);
    X3D_CHECK_ARG((mx & ~(X3D_MX_GA | X3D_MX_X)) == 0);
    X3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    WgArgs A = {};
    const int Ho = strideHW == 2 ? (H - 1) / 2 + 1 : H, Wo = strideHW == 2 ? (W - 1) / 2 + 1 : W;
    A.g = (const float*)g; A.a = (const float*)a; A.cb = cb; A.x = (const float*)x; A.pre = pre; A.pre_act = pre_act;
    A.wpartial = wpartial; A.ga_bf = (mx & X3D_MX_GA) != 0; A.x_bf = (mx & X3D_MX_X) != 0;
    A.N = N; A.Ci = Cin; A.Co = Cout; A.P = T * Ho * Wo; A.Pin = (long long)T * H * W;
    A.strided = strideHW == 2; A.T = T; A.H = H; A.W = W; A.Ho = Ho; A.Wo = Wo;
    A.units_per_sample = cdiv(A.P, WG_UNIT);
    int tiled;
    wgrad_plan(N, A.P, Cout, Cin, strideHW == 1 && (A.Pin % 4 == 0), &tiled, &A.groups, &A.cob, &A.cib, &A.ct_run,
               &A.it_run);
    X3D_CHECK_ARG(A.cob * A.cib <= 65535);
    dim3 grid(A.groups, A.cob * A.cib), block(256);
    if (mx && tiled && (A.strided || x3d_opt(X3D_OPT_WGRAD_F32))) {
        x3d_set_error("pw_bwd_weight: no mixed-storage kernel for Cin=%d Cout=%d P=%d stride=%d", Cin, Cout, A.P, strideHW);
        return X3D_EINVAL;
    }
    if (tiled) {
        // split-bf16 MFMA (3 products, ~1e-5 on dW) by default; X3D_WGRAD_F32 selects the exact fp32-MFMA kernel
        hipStream_t s3 = (hipStream_t)stream;
        const int ns = x3d_opt(X3D_OPT_BWD_TERMS) == 2 ? 2 : 3;
        x3d_note_kernel(x3d_opt(X3D_OPT_WGRAD_F32) ? "pw_wgrad2_kernel" : "pw_wgrad3_kernel");
        if (x3d_opt(X3D_OPT_WGRAD_F32)) {
            hipLaunchKernelGGL(pw_wgrad2_kernel, grid, block, 0, s3, A);
        } else if (A.strided) {
#define WG3S_GO(CO_, CI_)                                                                                      \
    do {                                                                                                       \
        if (ns == 2) hipLaunchKernelGGL((pw_wgrad3_kernel<CO_, CI_, true, false, 2>), grid, block, 0, s3, A);   \
        else hipLaunchKernelGGL((pw_wgrad3_kernel<CO_, CI_, true, false, 3>), grid, block, 0, s3, A);           \
    } while (0)
            if (wg3_co(Cout) == 128) { if (wg3_ci(Cin) == 64) WG3S_GO(128, 64); else WG3S_GO(128, 32); }
            else { if (wg3_ci(Cin) == 64) WG3S_GO(64, 64); else WG3S_GO(64, 32); }
#undef WG3S_GO
        } else {
#define WG3_GO2(CO_, CI_, NS_)                                                                                 \
    do {                                                                                                       \
        if (mx) hipLaunchKernelGGL((pw_wgrad3_kernel<CO_, CI_, false, true, NS_>), grid, block, 0, s3, A);      \
        else hipLaunchKernelGGL((pw_wgrad3_kernel<CO_, CI_, false, false, NS_>), grid, block, 0, s3, A);        \
    } while (0)
#define WG3_GO(CO_, CI_) do { if (ns == 2) WG3_GO2(CO_, CI_, 2); else WG3_GO2(CO_, CI_, 3); } while (0)
            if (wg3_co(Cout) == 128) { if (wg3_ci(Cin) == 64) WG3_GO(128, 64); else WG3_GO(128, 32); }
            else { if (wg3_ci(Cin) == 64) WG3_GO(64, 64); else WG3_GO(64, 32); }
#undef WG3_GO2
#undef WG3_GO
        }
        X3D_LAUNCH_CHECK();
        return X3D_OK;
    }
    const bool vec = (A.P % 4 == 0) && !A.strided && (A.Pin % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
#define WG_LAUNCH(CT_, IT_)                                                                                \
    do {                                                                                                   \
        if (mx) {                                                                                          \
            if (vec) hipLaunchKernelGGL((pw_wgrad_kernel<CT_, IT_, true, true>), grid, block, 0, s, A);     \
            else hipLaunchKernelGGL((pw_wgrad_kernel<CT_, IT_, false, true>), grid, block, 0, s, A);        \
        } else {                                                                                           \
            if (vec) hipLaunchKernelGGL((pw_wgrad_kernel<CT_, IT_, true, false>), grid, block, 0, s, A);    \
            else hipLaunchKernelGGL((pw_wgrad_kernel<CT_, IT_, false, false>), grid, block, 0, s, A);       \
        }                                                                                                  \
    } while (0)
    x3d_note_kernel("pw_wgrad_kernel");
    if (A.ct_run <= 2 && A.it_run <= 2) WG_LAUNCH(2, 2);
    else if (A.ct_run <= 2) WG_LAUNCH(2, 4);
    else if (A.it_run <= 2) WG_LAUNCH(4, 2);
    else WG_LAUNCH(4, 4);
#undef WG_LAUNCH
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" size_t x3d_wgrad_job_bytes(void) { return sizeof(X3DWgradJob); }

extern "C" int x3d_pw_bwd_weight_batch(const X3DWgradJob* jobs, int njobs, void* stream) {
    X3D_CHECK_ARG(jobs && njobs > 0);
    hipStream_t s = (hipStream_t)stream;
    const bool f32 = x3d_opt(X3D_OPT_WGRAD_F32) != 0;
    const int ns = x3d_opt(X3D_OPT_BWD_TERMS) == 2 ? 2 : 3;
    // variant id: bit 0 = CI 64, bit 1 = CO 128, bit 2 = gathered (strided) input; -1 = not a wgrad3 shape
    // + 10: the mixed-storage builds of the same variants.  Their storage flags are per job, so when ANY job of the call
    // has a bf16 tensor the whole call runs on them -- one launch per tile variant either way (splitting the batch by
    // storage type halves the workgroups per launch and pays a ramp and a tail twice)
    static thread_local WgBatch B[20];
    for (int v = 0; v < 20; ++v) { B[v].njobs = 0; B[v].wg0[0] = 0; }
    bool any_mx = false;
    for (int i = 0; i < njobs; ++i) any_mx = any_mx || jobs[i].mx != 0;
    auto launch = [&](int vv) -> int {
        WgBatch& b = B[vv];
        if (b.njobs == 0) return X3D_OK;
        const bool mxb = vv >= 10;
        const int v = vv % 10;
        const dim3 grid(b.wg0[b.njobs]), block(256);
        x3d_note_kernel(v >= 8 ? "pw_wgrad4_batch_kernel" : "pw_wgrad3_batch_kernel");
        if (v >= 8) {               // wide tiles: 8 waves, dynamic LDS (NS planes of dY and of the input)
            const size_t l1 = (size_t)ns * (256 + 96) * W3_LD * 2, l2 = (size_t)ns * (128 + 224) * W3_LD * 2;
#define WB4_GO(CO_, CI_, MX_, NS_, LDS_)                                                                                 \
    do {                                                                                                                \
        static bool attr_done = false;                                                                                  \
        if (!attr_done) {                                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad4_batch_kernel<CO_, CI_, MX_, NS_>),        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, NS_ * (CO_ + CI_) * W3_LD * 2);        \
            attr_done = true;                                                                                           \
        }                                                                                                               \
        hipLaunchKernelGGL((pw_wgrad4_batch_kernel<CO_, CI_, MX_, NS_>), grid, dim3(512), LDS_, s, b);                   \
    } while (0)
#define WB4_GO2(CO_, CI_, LDS_)                                                                                          \
    do {                                                                                                                \
        if (ns == 2) { if (mxb) WB4_GO(CO_, CI_, true, 2, LDS_); else WB4_GO(CO_, CI_, false, 2, LDS_); }                \
        else { if (mxb) WB4_GO(CO_, CI_, true, 3, LDS_); else WB4_GO(CO_, CI_, false, 3, LDS_); }                        \
    } while (0)
            if (v == 8) WB4_GO2(256, 96, l1); else WB4_GO2(128, 224, l2);
#undef WB4_GO2
#undef WB4_GO
            b.njobs = 0;
            X3D_LAUNCH_CHECK();
            return X3D_OK;
        }
#define WB_GO2(CO_, CI_, NS_)                                                                                     \
    do {                                                                                                        \
        if (mxb) hipLaunchKernelGGL((pw_wgrad3_batch_kernel<CO_, CI_, false, true, NS_>), grid, block, 0, s, b); \
        else hipLaunchKernelGGL((pw_wgrad3_batch_kernel<CO_, CI_, false, false, NS_>), grid, block, 0, s, b);    \
    } while (0)
#define WB_GO(CO_, CI_) do { if (ns == 2) WB_GO2(CO_, CI_, 2); else WB_GO2(CO_, CI_, 3); } while (0)
#define WBS_GO(CO_, CI_)                                                                                         \
    do {                                                                                                        \
        if (ns == 2) hipLaunchKernelGGL((pw_wgrad3_batch_kernel<CO_, CI_, true, false, 2>), grid, block, 0, s, b); \
        else hipLaunchKernelGGL((pw_wgrad3_batch_kernel<CO_, CI_, true, false, 3>), grid, block, 0, s, b);       \
    } while (0)
        switch (v) {
            case 0: WB_GO(64, 32); break;
            case 1: WB_GO(64, 64); break;
            case 2: WB_GO(128, 32); break;
            case 3: WB_GO(128, 64); break;
            case 4: WBS_GO(64, 32); break;
            case 5: WBS_GO(64, 64); break;
            case 6: WBS_GO(128, 32); break;
            default: WBS_GO(128, 64); break;
        }
#undef WBS_GO
#undef WB_GO2
#undef WB_GO
        b.njobs = 0;
        X3D_LAUNCH_CHECK();
        return X3D_OK;
    };
    for (int i = 0; i < njobs; ++i) {
        const X3DWgradJob& J = jobs[i];
        X3D_CHECK_ARG(J.g && J.a && J.cb && J.x && J.wpartial);
        X3D_CHECK_ARG(J.N > 0 && J.Cin > 0 && J.Cout > 0 && J.T > 0 && J.H > 0 && J.W > 0);
        X3D_CHECK_ARG(J.strideHW == 1 || J.strideHW == 2);
        WgArgs A = {};
        const int Ho = J.strideHW == 2 ? (J.H - 1) / 2 + 1 : J.H, Wo = J.strideHW == 2 ? (J.W - 1) / 2 + 1 : J.W;
        X3D_CHECK_ARG((J.mx & ~(X3D_MX_GA | X3D_MX_X)) == 0);
        A.g = (const float*)J.g; A.a = (const float*)J.a; A.cb = J.cb; A.x = (const float*)J.x; A.pre = J.pre;
        A.pre_act = J.pre_act; A.wpartial = J.wpartial; A.ga_bf = (J.mx & X3D_MX_GA) != 0; A.x_bf = (J.mx & X3D_MX_X) != 0;
        A.N = J.N; A.Ci = J.Cin; A.Co = J.Cout; A.P = J.T * Ho * Wo; A.Pin = (long long)J.T * J.H * J.W;
        A.strided = J.strideHW == 2; A.T = J.T; A.H = J.H; A.W = J.W; A.Ho = Ho; A.Wo = Wo;
        A.units_per_sample = cdiv(A.P, WG_UNIT);
        int tiled;
        wgrad_plan(J.N, A.P, J.Cout, J.Cin, J.strideHW == 1 && (A.Pin % 4 == 0), &tiled, &A.groups, &A.cob, &A.cib,
                   &A.ct_run, &A.it_run);
        if (!tiled || f32) {           // shapes outside the split-bf16 tiled kernel: one launch of their own
            const int rc = x3d_pw_bwd_weight(J.g, J.a, J.cb, J.x, J.pre, J.pre_act, J.wpartial, J.N, J.Cin, J.Cout, J.T,
                                             J.H, J.W, J.strideHW, J.mx, stream);
            if (rc != X3D_OK) return rc;
            continue;
        }
        int v = (wg3_ci(J.Cin) == 64 ? 1 : 0) | (wg3_co(J.Cout) == 128 ? 2 : 0) | (A.strided ? 4 : 0);
        const int wide = wg4_kind(J.Cout, J.Cin, !A.strided);
        if (wide) {                // same voxel groups (wpartial is sized by them), fewer and wider channel blocks
            v = 7 + wide;
            A.cob = cdiv(J.Cout, wide == 1 ? 256 : 128);
            A.cib = cdiv(J.Cin, wide == 1 ? 96 : 224);
        }
        if (J.mx && A.strided) {
            x3d_set_error("pw_bwd_weight_batch: no mixed-storage kernel for a strided conv (Cin=%d Cout=%d)", J.Cin, J.Cout);
            return X3D_EINVAL;
        }
        if (any_mx && !A.strided) v += 10;
        WgBatch& b = B[v];
        b.job[b.njobs] = A;
        b.wg0[b.njobs + 1] = b.wg0[b.njobs] + A.groups * A.cob * A.cib;
        if (++b.njobs == WB_MAX) {
            const int rc = launch(v);
            if (rc != X3D_OK) return rc;
        }
    }
    for (int v = 0; v < 20; ++v) {
        const int rc = launch(v);
        if (rc != X3D_OK) return rc;
    }
    return X3D_OK;
}

extern "C" int x3d_reduce_partials(const float* partial, float* out, int groups, int n, void* stream) {
    X3D_CHECK_ARG(partial && out && groups > 0 && n > 0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream,
                       partial, out, groups, n);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
