// Fused backward of a pointwise (1x1x1) convolution: data gradient AND weight gradient from ONE pass over the
// tensors (stages 1-2, where both are HBM bound and the weight gradient re-read exactly what the data gradient reads).
//
// Reference call sites replaced: the autograd backward (ConvolutionBackward: grad_input + grad_weight) of conv1x1x1
// (x3d.py:98-103) as Bottleneck.conv1 / conv3 (:112,116,146,162), fused with the BN backward in front of it
// (SubBatchNorm3d, x3d.py:47-58), the activation backward behind it (ReLU :148 / SwishEfficient.backward :80-84) or the
// residual-add + ReLU backward of the producing block (:165-169).
//
//   dY[co,p]  = cb0*g + cb1*a + cb2                                   (BN backward, coefficients per (sample, channel))
//   dX[ci,p]  = sum_co W[co,ci] dY[co,p]                              (data gradient; epilogue as in pw.hip)
//   dW[co,ci] = sum_{n,p} dY[co,p] X[ci,p],  X = act(pre0*x + pre1)   (weight gradient; per-workgroup partials)
//
// A workgroup walks a strided list of 64-voxel chunks.  Per chunk dY[CO][64] and X[CI][64] are staged ONCE into LDS
// as split-bf16 planes (v = hi + lo, 16 significant bits, 3 MFMA products hi*hi + hi*lo + lo*hi in fp32 -- the precision
// of the separate backward kernels) in [channel][voxel] order:
//   * weight gradient: voxel index on the MFMA K dimension, both operands read by rows (ds_read_b128);
//   * data gradient: channel index on K -- the same dY image read TRANSPOSED with ds_read_b64_tr_b16 (gfx950), so the
//     second layout costs no second copy; A operand = the pre-split transposed weight pack (x3d_pw_pack, L2 resident).
// Voxel v of a 32-voxel half chunk sits at column (v & 1) * 16 + (v >> 1): the two 16-column MFMA tiles of a half then give
// every lane two adjacent voxels (float2 stores, 128 B per row and wave instruction); the permutation is the same on
// both weight-gradient operands, so their contraction does not see it.
// dW stays in the accumulators for the workgroup's whole life; one partial per workgroup goes to HBM at the end.
#include <cstdlib>
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

enum { FE_PLAIN = 0, FE_ACTBWD = 1, FE_RESBWD = 2 };

struct FbArgs {
    const float* g; const float* a; const float* cb;     // [N][Co][P], [N][Co][P], [N][Co][3]
    const float* x; const float* xpre; int xact;         // [N][Ci][P]; [N][Ci][2] or NULL
    const float* wpt;                                    // transposed pack: fp32 image, then bf16 hi / lo planes
    float* dx;                                           // [N][Ci][P]
    float* wpartial;                                     // [groups][Co][Ci]
    float* partial;                                      // [N][Ci][tiles][2] (FE_ACTBWD / FE_RESBWD)
    const float* ex;                                     // FE_RESBWD: raw conv3 output of the producing block [N][Ci][P]
    const float* addend; int addend_stride;              // [N][Ci][P] (stride 1) or [N][Ci][T][Ho][Wo] (stride 2), or NULL
    int N, Co, Ci, P, tiles, T, H, W, Ho, Wo;
};

constexpr int F_LD = 72;           // bf16 elements per LDS row (144 B: conflict-free b128 row reads)
constexpr int F_PT = 64;           // voxels per chunk

__device__ __forceinline__ bf16x8 cat8(bf16x4 lo4, bf16x4 hi4) {
    return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// CO: dY channels padded to a multiple of 32 (<= 128), CI: input channels padded to a multiple of 32 (<= 128).
template <int CO, int CI, int EPI>
__global__ __launch_bounds__(256, (CO * CI >= 8192) ? 1 : 2) void pw_bwd_fused_kernel(const FbArgs A) {
    constexpr int ND = CO / 16, NX = CI / 16;              // staged float4 slots per thread (dY rows, X rows)
    constexpr int KS = CO / 32;                            // k steps (32 channels) of the data gradient
    constexpr int U = CI / 32;                             // data-gradient units per wave: ci tiles (wave >> 1) + 2 j
    constexpr bool BYCO = (CO / 16) >= 4;                  // weight-gradient tiles: waves split the co tiles, or the ci tiles
    constexpr int MW = BYCO ? CO / 64 : CO / 16;
    constexpr int NW = BYCO ? CI / 16 : CI / 64;
    static_assert(CO % 32 == 0 && CI % 32 == 0 && (BYCO || CI % 64 == 0), "tile shape");

    __shared__ __attribute__((aligned(16))) __bf16 Dh[CO * F_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Dlo[CO * F_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xh[CI * F_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xlo[CI * F_LD];
    __shared__ float red[(EPI == FE_PLAIN) ? 4 : 4 * U * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int half = wave & 1, mpar = wave >> 1;
    const int P = A.P, Co = A.Co, Ci = A.Ci;
    const int cps = (P + F_PT - 1) / F_PT;                 // chunks per sample
    const int total = A.N * cps;
    const int G = gridDim.x;

    // ---- staging roles: rows row0 + 16 i, voxels c4 .. c4 + 3 of the chunk (256 B contiguous per row and 16 lanes)
    const int c4 = (tid & 15) * 4;
    const int row0 = tid >> 4;
    // LDS column of voxel c4 (even) / c4 + 1 (odd); c4 + 2 / c4 + 3 follow at + 1
    const int colE = (c4 & 32) + ((c4 & 31) >> 1), colO = colE + 16;

    float4 rg[ND], ra[ND], rx[NX];
    float k0[ND], k1[ND], k2[ND], sc[NX], sh[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { sc[i] = 1.f; sh[i] = 0.f; }

    auto fetch = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * F_PT;
        const int pc = min(pt + c4, P - 4);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int co = min(row0 + 16 * i, Co - 1);
            const size_t base = ((size_t)n * Co + co) * (size_t)P + pc;
            rg[i] = *reinterpret_cast<const float4*>(A.g + base);
            ra[i] = *reinterpret_cast<const float4*>(A.a + base);
            const float* cb = A.cb + ((size_t)n * Co + co) * 3;
            k0[i] = cb[0]; k1[i] = cb[1]; k2[i] = cb[2];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int ci = min(row0 + 16 * i, Ci - 1);
            rx[i] = *reinterpret_cast<const float4*>(A.x + ((size_t)n * Ci + ci) * (size_t)P + pc);
            if (A.xpre != nullptr) {
                const float2 p2 = *reinterpret_cast<const float2*>(A.xpre + ((size_t)n * Ci + ci) * 2);
                sc[i] = p2.x; sh[i] = p2.y;
            }
        }
    };

    auto put = [&](__bf16* plane_h, __bf16* plane_l, int row, const float (&v)[4]) {
        bf16x2 he, ho, le, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)v[e];
            const __bf16 l = (__bf16)(v[e] - (float)h);
            if (e & 1) { ho[e >> 1] = h; lo[e >> 1] = l; } else { he[e >> 1] = h; le[e >> 1] = l; }
        }
        *reinterpret_cast<bf16x2*>(&plane_h[row * F_LD + colE]) = he;
        *reinterpret_cast<bf16x2*>(&plane_h[row * F_LD + colO]) = ho;
        *reinterpret_cast<bf16x2*>(&plane_l[row * F_LD + colE]) = le;
        *reinterpret_cast<bf16x2*>(&plane_l[row * F_LD + colO]) = lo;
    };

    auto store = [&](int c) {
        const int n = c / cps, pt = (c - n * cps) * F_PT;
        const bool pvv = pt + c4 < P;                       // P % 4 == 0: all four voxels or none
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = pvv && (row0 + 16 * i < Co);
            const float gv[4] = {rg[i].x, rg[i].y, rg[i].z, rg[i].w}, av[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? fmaf(k0[i], gv[e], fmaf(k1[i], av[e], k2[i])) : 0.f;
            put(Dh, Dlo, row0 + 16 * i, v);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool ok = pvv && (row0 + 16 * i < Ci);
            float v[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
            if (A.xpre != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(fmaf(sc[i], v[e], sh[i]), A.xact);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            put(Xh, Xlo, row0 + 16 * i, v);
        }
    };

    // ---- weight gradient: this wave's MW x NW tiles of dW, voxel index on K (two steps of 32 per chunk)
    f32x4 wacc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) wacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wco0 = BYCO ? MW * wave : 0, wci0 = BYCO ? 0 : NW * wave;          // first co / ci tile of this wave

    auto compute_w = [&]() {
#pragma unroll
        for (int s = 0; s < F_PT / 32; ++s) {
            bf16x8 ah[MW], al[MW], bh[NW], bl[NW];
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                const int off = ((wco0 + i) * 16 + r) * F_LD + s * 32 + 8 * q;
                ah[i] = *reinterpret_cast<const bf16x8*>(&Dh[off]);
                al[i] = *reinterpret_cast<const bf16x8*>(&Dlo[off]);
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int off = ((wci0 + j) * 16 + r) * F_LD + s * 32 + 8 * q;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Xh[off]);
                bl[j] = *reinterpret_cast<const bf16x8*>(&Xlo[off]);
            }
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], wacc[i][j], 0, 0, 0);
                    wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], wacc[i][j], 0, 0, 0);
                    wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], wacc[i][j], 0, 0, 0);
                }
        }
    };

    // ---- data gradient: units (ci tile (wave >> 1) + 2 j, 32-voxel half wave & 1), channel index on K
    const int mtiles = (Ci + 15) / 16, kg16 = (Co + 15) / 16, kg32 = (Co + 31) / 32;
    const __bf16* wqh = reinterpret_cast<const __bf16*>(A.wpt + (size_t)mtiles * kg16 * 256);
    const __bf16* wql = wqh + (size_t)mtiles * kg32 * 512;
    f32x4 dacc[U][2];

    // transposed fragment of the dY image: lane (col i = lane & 15, k group g = lane >> 4) gets dY[32 s + 8 g + 0..7][col]
    // for the 16 columns starting at col0; lane 4 qq + pp of a 16-lane group supplies the address of row qq, columns 4 pp ..
    const int tr_row = 8 * q + (r >> 2), tr_col = 4 * (r & 3);
    auto tr_frag = [&](const __bf16* plane, int s, int col0) -> bf16x8 {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = plane + (32 * s + tr_row) * F_LD + col0 + tr_col;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * F_LD));
        return cat8(v0, v1);
    };

    auto compute_d = [&]() {
#pragma unroll
        for (int j = 0; j < U; ++j) { dacc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dacc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                bh[h2] = tr_frag(Dh, s, 32 * half + 16 * h2);
                bl[h2] = tr_frag(Dlo, s, 32 * half + 16 * h2);
            }
            const int sc_ = min(s, kg32 - 1);               // clamped: the dY rows of a k step past Co are zero
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int mt = min(mpar + 2 * j, mtiles - 1);                  // clamped: duplicates are never stored
                const size_t off = (((size_t)mt * kg32 + sc_) * 64 + lane) * 8;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wqh + off);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(wql + off);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[h2], dacc[j][h2], 0, 0, 0);
                    dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[h2], dacc[j][h2], 0, 0, 0);
                    dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[h2], dacc[j][h2], 0, 0, 0);
                }
            }
        }
    };

    const bool has_add = A.addend != nullptr;
    const bool add_s2 = has_add && A.addend_stride == 2;
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;

    // epilogue of one chunk: lane owns rows 4 q + e of each unit and voxels 32 half + 2 r, + 1
    auto epilogue = [&](int c) {
        const int n = c / cps, tile = c - n * cps;
        const int pl = tile * F_PT + 32 * half + 2 * r;
        const bool pv = pl < P;                               // P even: both voxels or none
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
            // phase 1: the four rows' reads, branch-free from clamped addresses (the x rows were fetched by this
            // workgroup for the staging of this very chunk: L2 hits)
            float xv[4][2], ev[4][2], adv[4][2], esc[4], esh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = lt * 16 + 4 * q + e;
                const size_t mrow = (size_t)n * Ci + (ml < Ci ? ml : 0);
                if (EPI == FE_ACTBWD) {
                    const float2 c2 = *reinterpret_cast<const float2*>(A.xpre + mrow * 2);
                    esc[e] = c2.x; esh[e] = c2.y;
                }
                if (EPI != FE_PLAIN) {
                    const float2 t2 = *reinterpret_cast<const float2*>(A.x + mrow * (size_t)P + pc);
                    xv[e][0] = t2.x; xv[e][1] = t2.y;
                }
                if (EPI == FE_RESBWD) {
                    const float2 t2 = *reinterpret_cast<const float2*>(A.ex + mrow * (size_t)P + pc);
                    ev[e][0] = t2.x; ev[e][1] = t2.y;
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
                const bool mv = ml < Ci;
                float v[2] = {dacc[j][0][e], dacc[j][1][e]};
                float s1 = 0.f, s2 = 0.f;
                if (has_add) { v[0] += av[0] ? adv[e][0] : 0.f; v[1] += av[1] ? adv[e][1] : 0.f; }
                if (EPI != FE_PLAIN) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const float xj = pv ? xv[e][j2] : 0.f;
                        float mul;                                   // statistics multiplier
                        if (EPI == FE_RESBWD) {
                            v[j2] = (pv && xj > 0.f) ? v[j2] : 0.f;
                            mul = pv ? ev[e][j2] : 0.f;
                        } else {
                            v[j2] = pv ? v[j2] * act_bwd(fmaf(esc[e], xj, esh[e]), A.xact) : 0.f;
                            mul = xj;
                        }
                        s1 += v[j2];
                        s2 = fmaf(v[j2], mul, s2);
                    }
                }
                if (mv && pv)
                    *reinterpret_cast<float2*>(A.dx + ((size_t)n * Ci + ml) * (size_t)P + pl) = make_float2(v[0], v[1]);
                if (EPI != FE_PLAIN) {
                    s1 = row16_sum(s1);
                    s2 = row16_sum(s2);
                    if (r == 0) {
                        red[((wave * U + j) * 16 + 4 * q + e) * 2] = mv ? s1 : 0.f;
                        red[((wave * U + j) * 16 + 4 * q + e) * 2 + 1] = mv ? s2 : 0.f;
                    }
                }
            }
        }
    };

    // one partial per (row, 64-voxel tile): the two half-chunk waves of a row are summed (after the chunk's barrier)
    auto write_stats = [&](int c) {
        const int n = c / cps, tile = c - n * cps;
        for (int idx = tid; idx < Ci * 2; idx += 256) {
            const int ml = idx >> 1, which = idx & 1;
            const int lt = ml >> 4, wv = (lt & 1) * 2, j = lt >> 1;
            const float s = red[((wv * U + j) * 16 + (ml & 15)) * 2 + which] +
                            red[(((wv + 1) * U + j) * 16 + (ml & 15)) * 2 + which];
            A.partial[(((size_t)n * Ci + ml) * A.tiles + tile) * 2 + which] = s;
        }
    };

    int c = blockIdx.x;
    if (c < total) {
        fetch(c);
        for (; c < total; c += G) {
            store(c);
            __syncthreads();                 // chunk staged
            const int cn = c + G;
            if (cn < total) fetch(cn);
            compute_w();
            compute_d();
            epilogue(c);
            __syncthreads();                 // everyone done reading this chunk's images; red[] complete
            if (EPI != FE_PLAIN) write_stats(c);
        }
    }

    // D[i = co][j = ci]: lane (q, r), reg e -> co = tile * 16 + 4 q + e, ci = tile * 16 + r
    float* out = A.wpartial + (size_t)blockIdx.x * Co * Ci;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int oc = (wco0 + i) * 16 + 4 * q + e, ic = (wci0 + j) * 16 + r;
                if (oc < Co && ic < Ci) out[(size_t)oc * Ci + ic] = wacc[i][j][e];
            }
}

static int fb_pad32(int c) { return (c + 31) / 32 * 32; }

}  // namespace

// shapes the fused kernel is built for: dense, P % 4 == 0, padded (Co, Ci) in the instantiated set
extern "C" int x3d_pw_bwd_fused_ok(int Cin, int Cout, int P) {
    if (P % 4 != 0 || P < 4) return 0;
    const int co = fb_pad32(Cout), ci = fb_pad32(Cin) == 96 ? 128 : fb_pad32(Cin);
    const int cop = co == 96 ? 128 : co;
    if (cop > 128 || ci > 128) return 0;
    if (cop == 32 && ci != 64) return 0;           // instantiated: (32,64) (64,32) (64,64) (64,128) (128,32) (128,64)
    if (cop == 64 && !(ci == 32 || ci == 64 || ci == 128)) return 0;
    if (cop == 128 && !(ci == 32 || ci == 64)) return 0;
    return 1;
}

extern "C" int x3d_pw_bwd_fused_groups(int N, int P) {
    const long long chunks = (long long)N * ((P + F_PT - 1) / F_PT);
    static const int gmax = getenv("X3D_FB_GRID") ? atoi(getenv("X3D_FB_GRID")) : 512;
    return (int)(chunks < gmax ? chunks : gmax);
}

extern "C" int x3d_pw_bwd_fused_tiles(int P) { return (P + F_PT - 1) / F_PT; }

// mode: 0 = plain (+ addend), 1 = activation backward (x raw, xpre, xact; statistics {sum out, sum out * x}),
//       2 = residual-add + ReLU backward of the producing block (x = its output, ex = its raw conv3 output)
extern "C" int x3d_pw_bwd_fused(const float* g, const float* a, const float* cb, const float* wpacked_t, const float* x,
                                const float* xpre, int xact, int mode, const float* ex, const float* addend,
                                int addend_stride, float* dx, float* wpartial, float* partial, int N, int Cin, int Cout,
                                int T, int H, int W, void* stream) {
    X3D_CHECK_ARG(g && a && cb && wpacked_t && x && dx && wpartial);
    X3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(mode >= 0 && mode <= 2 && (addend_stride == 1 || addend_stride == 2));
    X3D_CHECK_ARG(mode == 0 || partial != nullptr);
    X3D_CHECK_ARG(mode != 1 || xpre != nullptr);
    X3D_CHECK_ARG(mode != 2 || (ex != nullptr && xpre == nullptr));
    const int P = T * H * W;
    if (!x3d_pw_bwd_fused_ok(Cin, Cout, P)) {
        x3d_set_error("x3d_pw_bwd_fused: shape Cin=%d Cout=%d P=%d is outside the fused kernel's set", Cin, Cout, P);
        return X3D_EINVAL;
    }
    FbArgs A = {};
    A.g = g; A.a = a; A.cb = cb; A.x = x; A.xpre = xpre; A.xact = xact; A.wpt = wpacked_t; A.dx = dx;
    A.wpartial = wpartial; A.partial = partial; A.ex = ex; A.addend = addend; A.addend_stride = addend_stride;
    A.N = N; A.Co = Cout; A.Ci = Cin; A.P = P; A.tiles = (P + F_PT - 1) / F_PT; A.T = T; A.H = H; A.W = W;
    A.Ho = (H - 1) / 2 + 1; A.Wo = (W - 1) / 2 + 1;
    const dim3 grid(x3d_pw_bwd_fused_groups(N, P)), block(256);
    hipStream_t s = (hipStream_t)stream;
    int co = fb_pad32(Cout), ci = fb_pad32(Cin);
    if (co == 96) co = 128;
    if (ci == 96) ci = 128;
#define FB_LAUNCH(CO_, CI_)                                                                                     \
    do {                                                                                                        \
        if (mode == 0) hipLaunchKernelGGL((pw_bwd_fused_kernel<CO_, CI_, FE_PLAIN>), grid, block, 0, s, A);      \
        else if (mode == 1) hipLaunchKernelGGL((pw_bwd_fused_kernel<CO_, CI_, FE_ACTBWD>), grid, block, 0, s, A); \
        else hipLaunchKernelGGL((pw_bwd_fused_kernel<CO_, CI_, FE_RESBWD>), grid, block, 0, s, A);               \
    } while (0)
    if (co == 32 && ci == 64) FB_LAUNCH(32, 64);
    else if (co == 64 && ci == 32) FB_LAUNCH(64, 32);
    else if (co == 64 && ci == 64) FB_LAUNCH(64, 64);
    else if (co == 64 && ci == 128) FB_LAUNCH(64, 128);
    else if (co == 128 && ci == 32) FB_LAUNCH(128, 32);
    else FB_LAUNCH(128, 64);
#undef FB_LAUNCH
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
