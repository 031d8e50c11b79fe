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
    const void* g; const void* a; const float* cb;       // [N][Co][P], [N][Co][P] (bf16 in the MX = 2 build), [N][Co][3]
    const void* x; const float* xpre; int xact;          // [N][Ci][P] (bf16 in the MX = 1 build); [N][Ci][2] or NULL
    const float* wpt;                                    // transposed pack: fp32 image, then bf16 hi / lo planes
    void* dx;                                            // [N][Ci][P] (bf16 in the MX = 1 build)
    float* wpartial;                                     // [groups][Co][Ci]
    float* partial;                                      // [N][Ci][tiles][2] (FE_ACTBWD / FE_RESBWD)
    const float* ex;                                     // FE_RESBWD: raw conv3 output of the producing block [N][Ci][P]
    const float* addend; int addend_stride;              // [N][Ci][P] (stride 1) or [N][Ci][T][Ho][Wo] (stride 2), or NULL
    int N, Co, Ci, P, tiles, T, H, W, Ho, Wo;
};

constexpr int F_LD = 72;           // bf16 elements per LDS row (144 B: conflict-free b128 row reads)
constexpr int F_PT = 64;           // voxels per chunk
constexpr int F_RL = 68;           // floats per row of the raw fp32 LDS tiles

// load from a wave-uniform base (SGPR pair) + 32-bit byte offset (one VGPR): `global_load ... v_off, s[base]`
template <typename T>
__device__ __forceinline__ T ldg_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ bf16x8 cat8(bf16x4 lo4, bf16x4 hi4) {
    return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// CO: dY channels padded to a multiple of 32 (<= 128), CI: input channels padded to a multiple of 32 (<= 128).
// NWV waves per workgroup (4 or 8); OCC = waves per SIMD the register budget is sized for.
// Occupancy is the latency hiding here: a workgroup runs its chunk as fetch -> stage -> MFMA -> epilogue without a register
// prefetch across chunks (that version needed 180-256 registers and ran at 1-2 waves per SIMD, 2x off the HBM time);
// the global round trips of one workgroup are covered by the other 3-4 resident on the CU.
// MX: mixed-storage build (BASELINE config 5): 0 = every tensor fp32; 1 = x and dx are bf16 (conv3 of a bottleneck: the wide
// Cmid tensors are its input side); 2 = g and a are bf16 (conv1: the wide tensors are its output side).
// NS: bf16 terms per fp32 operand of both GEMMs -- 3 (hi + mid + lo = all 24 significant bits, six MFMA products per tile
// step: fp32-level accuracy, the default) or 2 (hi + mid, three products, ~2^-16 per product: option bwd_terms = 2).

// rows of the raw fp32 epilogue tiles (x / ex / addend).  The widest shape (128 x 64: conv1 of stage 2, 48 input channels
// in X3D-S/M/L) keeps 48 -- what lets all three weight planes of that shape stay in LDS; x3d_pw_bwd_fused_ok refuses
// more input channels there
constexpr int fb_raw_rows(int CO, int CI) { return (CO == 128 && CI == 64) ? 48 : CI; }

// LDS bytes of one instantiation with WP weight planes in LDS (the kernel's __shared__ arrays, in declaration order)
constexpr int fb_lds_bytes(int CO, int CI, int EPI, bool ADD, int NWV, int NS, int WP) {
    const int MP = NWV / 2, U = (CI / 16 + MP - 1) / MP, KS = CO / 32, CR = fb_raw_rows(CO, CI);
    return NS * (CO + CI) * F_LD * 2 + ((EPI == FE_PLAIN) ? 4 : NWV * U * 16 * 2) * 4 + (CI / 16) * KS * WP * 64 * 8 * 2 +
           ((EPI == FE_ACTBWD) ? CR * F_RL : 4) * 4 + ((EPI == FE_RESBWD) ? CR * F_RL : 4) * 4 + (ADD ? CR * F_RL : 4) * 4 +
           ((EPI == FE_ACTBWD) ? CI : 1) * 8;
}
constexpr int FB_LDS_MAX = 160 * 1024;
// is (CO, CI, epilogue, addend) built for NS terms?  (the lo weight plane may move to registers, nothing else gives)
constexpr bool fb_fits(int CO, int CI, int EPI, bool ADD, int NWV, int NS) {
    return fb_lds_bytes(CO, CI, EPI, ADD, NWV, NS, 2) <= FB_LDS_MAX;
}

template <int CO, int CI, int EPI, bool ADD, int NWV, int OCC, int MX, int NS>
__global__ __launch_bounds__(64 * NWV, OCC) void pw_bwd_fused_kernel(const FbArgs A) {
    constexpr int GA_BF = MX == 2, X_BF = MX == 1, DX_BF = MX == 1;
    // weight planes kept in LDS: all NS of them when that fits, else hi + mid with the lo plane's fragments held in
    // registers for the workgroup's life (the two widest stage-2 shapes; loaded once, before the chunk loop)
    constexpr int WP = fb_lds_bytes(CO, CI, EPI, ADD, NWV, NS, NS) <= FB_LDS_MAX ? NS : 2;
    constexpr int CR = fb_raw_rows(CO, CI);                // rows of the raw epilogue tiles (>= Ci: checked by the host)
    static_assert(fb_fits(CO, CI, EPI, ADD, NWV, NS), "LDS budget: x3d_pw_bwd_fused_ok must refuse this combination");
    constexpr int NT = 64 * NWV;                           // threads
    constexpr int RP = NT / 16;                            // rows staged per pass (16 lanes x float4 = one 64-voxel row)
    constexpr int ND = (CO + RP - 1) / RP, NX = (CI + RP - 1) / RP;   // staged float4 slots per thread (dY rows, X rows)
    constexpr int KS = CO / 32;                            // k steps (32 channels) of the data gradient
    constexpr int MP = NWV / 2;                            // data-gradient: waves per voxel half
    constexpr int U = (CI / 16 + MP - 1) / MP;             // units per wave: ci tiles (wave >> 1) + MP j
    // weight gradient: WCO waves along the co tiles, WCI along the ci tiles
    constexpr int WCO = (CO / 16) < NWV ? (CO / 16) : NWV;
    constexpr int WCI = NWV / WCO;
    constexpr int MW = CO / 16 / WCO, NW = CI / 16 / WCI;
    static_assert(CO % 32 == 0 && CI % 32 == 0, "tile shape");
    static_assert(WCO * WCI == NWV && MW * WCO * 16 == CO && NW * WCI * 16 == CI, "weight-gradient tiling");

    __shared__ __attribute__((aligned(16))) __bf16 Dp[NS * CO * F_LD];          // dY planes: hi, mid(, lo)
    __shared__ __attribute__((aligned(16))) __bf16 Xp[NS * CI * F_LD];          // X planes
    __bf16* const Dh = Dp; __bf16* const Dm = Dp + CO * F_LD; __bf16* const Dlo = Dp + (NS - 1) * CO * F_LD;
    __bf16* const Xh = Xp; __bf16* const Xm = Xp + CI * F_LD; __bf16* const Xlo = Xp + (NS - 1) * CI * F_LD;
    __shared__ float red[(EPI == FE_PLAIN) ? 4 : NWV * U * 16 * 2];
    __shared__ __attribute__((aligned(16))) __bf16 Wl[(CI / 16) * KS * WP * 64 * 8];
    // epilogue operands of the chunk, staged raw (fp32, natural voxel order) with the same coalesced row loads as the
    // GEMM operands: x (activation backward + statistics; the residual mode takes its ReLU mask from the sign of the hi
    // plane of X instead -- x > 0 exactly when bf16(x) > 0 for every normal x -- and stages no raw x), ex, dense addend
    __shared__ __attribute__((aligned(16))) float Xr[(EPI == FE_ACTBWD) ? CR * F_RL : 4];
    __shared__ __attribute__((aligned(16))) float Er[(EPI == FE_RESBWD) ? CR * F_RL : 4];
    __shared__ __attribute__((aligned(16))) float Ar[ADD ? CR * F_RL : 4];
    __shared__ float2 Cf[(EPI == FE_ACTBWD) ? CI : 1];     // per-row BN coefficients of the chunk's sample (activation backward)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int half = wave & 1, mpar = wave >> 1;
    const int P = A.P, Co = A.Co, Ci = A.Ci;
    const int cps = (P + F_PT - 1) / F_PT;                 // chunks per sample
    const int total = A.N * cps;
    const int G = gridDim.x;

    // ---- staging roles: rows row0 + RP i, voxels c4 .. c4 + 3 of the chunk (256 B contiguous per row and 16 lanes)
    const int c4 = (tid & 15) * 4;
    const int row0 = tid >> 4;
    // LDS column of voxel c4 (even) / c4 + 1 (odd); c4 + 2 / c4 + 3 follow at + 1
    const int colE = (c4 & 32) + ((c4 & 31) >> 1), colO = colE + 16;

    auto put = [&](__bf16* plane_h, __bf16* plane_m, __bf16* plane_l, int row, const float (&v)[4]) {
        unsigned hE, mE, lE, hO, mO, lO;               // pairs (0, 2) and (1, 3): x3d_split3_pair, common.h
        x3d_split3_pair(v[0], v[2], hE, mE, lE);
        x3d_split3_pair(v[1], v[3], hO, mO, lO);
        const bf16x2 he = __builtin_bit_cast(bf16x2, hE), ho = __builtin_bit_cast(bf16x2, hO);
        const bf16x2 me = __builtin_bit_cast(bf16x2, mE), mo = __builtin_bit_cast(bf16x2, mO);
        const bf16x2 le = __builtin_bit_cast(bf16x2, lE), lo = __builtin_bit_cast(bf16x2, lO);
        *reinterpret_cast<bf16x2*>(&plane_h[row * F_LD + colE]) = he;
        *reinterpret_cast<bf16x2*>(&plane_h[row * F_LD + colO]) = ho;
        *reinterpret_cast<bf16x2*>(&plane_m[row * F_LD + colE]) = me;
        *reinterpret_cast<bf16x2*>(&plane_m[row * F_LD + colO]) = mo;
        if (NS == 3) {
            *reinterpret_cast<bf16x2*>(&plane_l[row * F_LD + colE]) = le;
            *reinterpret_cast<bf16x2*>(&plane_l[row * F_LD + colO]) = lo;
        }
    };

    // global -> registers (fetch, issued one chunk ahead) -> (BN-backward combine | forward prologue) -> split-bf16 LDS images
    float4 rg[ND], ra[ND], rx[NX], rex[(EPI == FE_RESBWD) ? NX : 1], radd[ADD ? NX : 1];
    float k0[ND], k1[ND], k2[ND], sc[NX], sh[NX];
    const bool add_dense = ADD && A.addend_stride == 1;
    unsigned add_even = 0;                                  // stride-2 addend: bit e = voxel e of this thread's four lands on the coarse grid
    auto fetch = [&](int n, int pt) {
        const int pc = min(pt + c4, P - 4);
        // stride-2 addend (the downsample branch's compact gradient [T][Ho][Wo], first block of a stage): its value for
        // voxel (t, h, w) exists where h and w are even; the offsets depend on the voxel only, so they are walked once
        unsigned aoff[4] = {0, 0, 0, 0};
        if (ADD && !add_dense) {
            const int hw = A.H * A.W;
            int t = pc / hw, rem = pc - t * hw;
            int h = rem / A.W, w = rem - h * A.W;
            add_even = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool even = !(h & 1) && !(w & 1);
                add_even |= even ? (1u << e) : 0u;
                aoff[e] = even ? (unsigned)((t * A.Ho + (h >> 1)) * A.Wo + (w >> 1)) : 0u;
                if (++w == A.W) { w = 0; if (++h == A.H) { h = 0; ++t; } }
            }
        }
        // per-sample bases are wave uniform (SGPRs); a row of one sample is < 4 GB away: 32-bit byte offsets
        const char* gs = mx_base(A.g, (size_t)n * Co * (size_t)P, GA_BF);
        const char* as = mx_base(A.a, (size_t)n * Co * (size_t)P, GA_BF);
        const char* xs = mx_base(A.x, (size_t)n * Ci * (size_t)P, X_BF);
        const float* cbs = A.cb + (size_t)n * Co * 3;
        const float* xps = A.xpre != nullptr ? A.xpre + (size_t)n * Ci * 2 : nullptr;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const unsigned co = (unsigned)min(row0 + RP * i, Co - 1);
            const unsigned off = co * (unsigned)P + (unsigned)pc;
            rg[i] = ldo4(gs, off, GA_BF);
            ra[i] = ldo4(as, off, GA_BF);
            k0[i] = ldg_off<float>(cbs, co * 12u); k1[i] = ldg_off<float>(cbs, co * 12u + 4u); k2[i] = ldg_off<float>(cbs, co * 12u + 8u);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const unsigned ci = (unsigned)min(row0 + RP * i, Ci - 1);
            const unsigned xoff = ci * (unsigned)P + (unsigned)pc;
            rx[i] = ldo4(xs, xoff, X_BF);
            if (EPI == FE_RESBWD) rex[i] = ldg_off<float4>(A.ex + (size_t)n * Ci * (size_t)P, xoff * 4u);
            if (ADD) {
                if (add_dense) {
                    radd[i] = ldg_off<float4>(A.addend + (size_t)n * Ci * (size_t)P, xoff * 4u);
                } else {
                    const unsigned addP = (unsigned)(A.T * A.Ho * A.Wo);
                    const float* ads = A.addend + (size_t)n * Ci * (size_t)addP;
                    const unsigned ab = ci * addP;
                    radd[i] = make_float4(ldg_off<float>(ads, (ab + aoff[0]) * 4u), ldg_off<float>(ads, (ab + aoff[1]) * 4u),
                                          ldg_off<float>(ads, (ab + aoff[2]) * 4u), ldg_off<float>(ads, (ab + aoff[3]) * 4u));
                }
            }
            sc[i] = 1.f; sh[i] = 0.f;
            if (xps != nullptr) {
                const float2 p2 = ldg_off<float2>(xps, ci * 8u);
                sc[i] = p2.x; sh[i] = p2.y;
            }
        }
    };
    auto store = [&](int pt) {
        const bool pvv = pt + c4 < P;                       // P % 4 == 0: all four voxels or none
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = pvv && (row0 + RP * i < Co);
            const float gv[4] = {rg[i].x, rg[i].y, rg[i].z, rg[i].w}, av[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? fmaf(k0[i], gv[e], fmaf(k1[i], av[e], k2[i])) : 0.f;
            if (CO % RP == 0 || row0 + RP * i < CO) put(Dh, Dm, Dlo, row0 + RP * i, v);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool ok = pvv && (row0 + RP * i < Ci);
            float v[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
            if (EPI == FE_ACTBWD && c4 == 0 && (CI % RP == 0 || row0 + RP * i < CI)) Cf[row0 + RP * i] = make_float2(sc[i], sh[i]);
            if ((CR % RP == 0 && CR == CI) || row0 + RP * i < CR) {
                if (EPI == FE_ACTBWD) *reinterpret_cast<float4*>(&Xr[(row0 + RP * i) * F_RL + c4]) = rx[i];
                if (EPI == FE_RESBWD) *reinterpret_cast<float4*>(&Er[(row0 + RP * i) * F_RL + c4]) = rex[i];
                if (ADD) {
                    float4 av4 = radd[i];
                    if (!add_dense) {
                        av4.x = (add_even & 1u) ? av4.x : 0.f; av4.y = (add_even & 2u) ? av4.y : 0.f;
                        av4.z = (add_even & 4u) ? av4.z : 0.f; av4.w = (add_even & 8u) ? av4.w : 0.f;
                    }
                    *reinterpret_cast<float4*>(&Ar[(row0 + RP * i) * F_RL + c4]) = av4;
                }
            }
            if (A.xpre != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(fmaf(sc[i], v[e], sh[i]), A.xact);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            if (CI % RP == 0 || row0 + RP * i < CI) put(Xh, Xm, Xlo, row0 + RP * i, v);
        }
    };

    // ---- weight gradient: this wave's MW x NW tiles of dW, voxel index on K (two steps of 32 per chunk)
    f32x4 wacc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) wacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wco0 = MW * (wave % WCO), wci0 = NW * (wave / WCO);          // first co / ci tile of this wave

    auto compute_w = [&]() {
#pragma unroll
        for (int s = 0; s < F_PT / 32; ++s) {
            bf16x8 ah[MW], am[MW], al[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                const int off = ((wco0 + i) * 16 + r) * F_LD + s * 32 + 8 * q;
                ah[i] = *reinterpret_cast<const bf16x8*>(&Dh[off]);
                am[i] = *reinterpret_cast<const bf16x8*>(&Dm[off]);
                if (NS == 3) al[i] = *reinterpret_cast<const bf16x8*>(&Dlo[off]);
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int off = ((wci0 + j) * 16 + r) * F_LD + s * 32 + 8 * q;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Xh[off]);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&Xm[off]);
                bf16x8 bl;
                if (NS == 3) bl = *reinterpret_cast<const bf16x8*>(&Xlo[off]);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (NS == 3) {                     // smallest terms first
                        wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, wacc[i][j], 0, 0, 0);
                        wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, wacc[i][j], 0, 0, 0);
                        wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm, wacc[i][j], 0, 0, 0);
                    }
                    wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh, wacc[i][j], 0, 0, 0);
                    wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm, wacc[i][j], 0, 0, 0);
                    wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, wacc[i][j], 0, 0, 0);
                }
            }
        }
    };

    // ---- data gradient: units (ci tile (wave >> 1) + MP j, 32-voxel half wave & 1), channel index on K
    const int mtiles = (Ci + 15) / 16, kg16 = (Co + 15) / 16, kg32 = (Co + 31) / 32;
    const __bf16* wqh = reinterpret_cast<const __bf16*>(A.wpt + (size_t)mtiles * kg16 * 256);
    const size_t wplane = (size_t)mtiles * kg32 * 512;                         // pack planes: hi, mid, lo
    f32x4 dacc[U][2];
    // A operand (transposed weights, pre-split, MFMA fragment order): chunk invariant -- copied once into LDS
    // ([ci tile][k step][plane][lane][8]), one ds_read_b128 per fragment and chunk instead of 8 registers per fragment
    for (int i = tid; i < (CI / 16) * KS * WP * 64; i += NT) {
        const int ln = i & 63, rest0 = i >> 6;
        const int pl = rest0 % WP, rest = rest0 / WP;
        const int s = rest % KS, mt = rest / KS;
        const int mtc = min(mt, mtiles - 1), sc_ = min(s, kg32 - 1);          // clamped: never stored / zero dY rows
        const size_t off = (((size_t)mtc * kg32 + sc_) * 64 + ln) * 8;
        *reinterpret_cast<bf16x8*>(&Wl[(size_t)i * 8]) = *reinterpret_cast<const bf16x8*>(wqh + pl * wplane + off);
    }
    // the lo plane's fragments of this wave's units when they do not fit the LDS (WP < NS): registers, loaded once
    bf16x8 alo[(WP < NS) ? U : 1][(WP < NS) ? KS : 1];
    if (WP < NS) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int mtc = min(min(mpar + MP * j, CI / 16 - 1), mtiles - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int sc_ = min(s, kg32 - 1);
                alo[j][s] = *reinterpret_cast<const bf16x8*>(wqh + 2 * wplane + (((size_t)mtc * kg32 + sc_) * 64 + lane) * 8);
            }
        }
    }

    // transposed fragment of the dY image: lane (col i = lane & 15, k group g = lane >> 4) gets dY[32 s + 8 g + 0..7][col]
    // for the 16 columns starting at col0; lane 4 qq + pp of a 16-lane group supplies the address of row qq, columns 4 pp ..
    const int tr_off = (8 * q + (r >> 2)) * F_LD + 4 * (r & 3) + 32 * half;
    auto tr_frag = [&](const __bf16* plane, int s, int h2) -> bf16x8 {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = plane + 32 * s * F_LD + tr_off + 16 * h2;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * F_LD));
        return cat8(v0, v1);
    };

    auto compute_d = [&]() {
#pragma unroll
        for (int j = 0; j < U; ++j) { dacc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dacc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (NS == 2) {
                bf16x8 bh[2], bm[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) { bh[h2] = tr_frag(Dh, s, h2); bm[h2] = tr_frag(Dm, s, h2); }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int mtl_ = min(mpar + MP * j, CI / 16 - 1);
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Wl[(((mtl_ * KS + s) * WP + 0) * 64 + lane) * 8]);
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(&Wl[(((mtl_ * KS + s) * WP + 1) * 64 + lane) * 8]);
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        // (round 4: dY fragment as the A operand -> transposed accumulator tile, see the epilogue)
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[h2], am, dacc[j][h2], 0, 0, 0);
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[h2], ah, dacc[j][h2], 0, 0, 0);
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[h2], ah, dacc[j][h2], 0, 0, 0);
                    }
                }
            } else {
                // three terms: one 16-voxel column tile at a time (three B fragments live instead of six; U is 1 for every
                // shape of the network, so the A fragments are read once per column tile)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const bf16x8 bh = tr_frag(Dh, s, h2), bm = tr_frag(Dm, s, h2), bl = tr_frag(Dlo, s, h2);
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const int mtl_ = min(mpar + MP * j, CI / 16 - 1);
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Wl[(((mtl_ * KS + s) * WP + 0) * 64 + lane) * 8]);
                        const bf16x8 am = *reinterpret_cast<const bf16x8*>(&Wl[(((mtl_ * KS + s) * WP + 1) * 64 + lane) * 8]);
                        bf16x8 al;
                        if (WP == 3) al = *reinterpret_cast<const bf16x8*>(&Wl[(((mtl_ * KS + s) * WP + 2) * 64 + lane) * 8]);
                        else al = alo[(WP < NS) ? j : 0][(WP < NS) ? s : 0];
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, dacc[j][h2], 0, 0, 0);      // smallest terms first
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, dacc[j][h2], 0, 0, 0);
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am, dacc[j][h2], 0, 0, 0);
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am, dacc[j][h2], 0, 0, 0);
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah, dacc[j][h2], 0, 0, 0);
                        dacc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, dacc[j][h2], 0, 0, 0);
                    }
                }
            }
        }
    };

    // epilogue of one chunk.  Round 4: the data-gradient accumulators are TRANSPOSED tiles (dY fragment = MFMA A operand): lane
    // (q, r) owns input channel lt * 16 + r and the 8 consecutive voxels 32 half + 8 q + jj = dacc[.][jj & 1][jj >> 1]
    // (voxel v of a half sits at LDS column (v & 1) * 16 + (v >> 1)) -- operands come out of the raw LDS tiles as two b128
    // reads each, dX goes out as two 16-byte stores, the statistics need two cross-row steps instead of 8 DPP adds per row.
    // Statistics (round 4): a workgroup owns a CONTIGUOUS chunk range, every lane accumulates ITS channel's two sums over its
    // voxels in registers (sa: two per unit -- the row-wise layout needed eight and spilled) and the four lane
    // groups of a channel are combined only when the range leaves the sample (`flush`): one partial per (sample, row,
    // workgroup) -- 65 instead of 784-1568 per row at stage 1 -- and no cross-lane step, LDS traffic or store per chunk.
    float sa[U][2];
#pragma unroll
    for (int j = 0; j < U; ++j) { sa[j][0] = 0.f; sa[j][1] = 0.f; }
    auto epilogue = [&](int n, int tile, bool flush) {
        const int vl = 32 * half + 8 * q;                     // first voxel within the chunk
        const int p0 = tile * F_PT + vl;
        const bool pva = p0 < P, pvb = p0 + 4 < P;            // P % 4 == 0: whole groups of four (clamped duplicates masked)
        char* dxs = mx_base(A.dx, (size_t)n * Ci * (size_t)P, DX_BF);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int lt = mpar + MP * j;
            if (lt * 16 >= CI) continue;                      // wave-uniform: this unit does not exist (16-wave variants)
            const int ml = lt * 16 + r;
            const bool mv = ml < Ci;
            const int mr = CR < CI ? min(ml, CR - 1) : ml;    // raw-tile row (rows >= Ci are masked by mv)
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = dacc[j][jj & 1][jj >> 1];
            if (ADD) {
                const float4 a0 = *reinterpret_cast<const float4*>(&Ar[mr * F_RL + vl]);
                const float4 a1 = *reinterpret_cast<const float4*>(&Ar[mr * F_RL + vl + 4]);
                v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
            }
            float s1 = 0.f, s2 = 0.f;
            if (EPI != FE_PLAIN) {
                float xa[8], ea[8];
                float esc = 1.f, esh = 0.f;
                if (EPI == FE_RESBWD) {
                    // ReLU mask of the producing block: sign of x from the hi plane of the staged X image (even voxels of
                    // the half at columns 4 q .. 4 q + 3, odd ones 16 further)
                    const bf16x4 xe = *reinterpret_cast<const bf16x4*>(&Xh[ml * F_LD + 32 * half + 4 * q]);
                    const bf16x4 xo = *reinterpret_cast<const bf16x4*>(&Xh[ml * F_LD + 32 * half + 4 * q + 16]);
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) { xa[2 * i2] = (float)xe[i2]; xa[2 * i2 + 1] = (float)xo[i2]; }
                    const float4 e0 = *reinterpret_cast<const float4*>(&Er[mr * F_RL + vl]);
                    const float4 e1 = *reinterpret_cast<const float4*>(&Er[mr * F_RL + vl + 4]);
                    ea[0] = e0.x; ea[1] = e0.y; ea[2] = e0.z; ea[3] = e0.w; ea[4] = e1.x; ea[5] = e1.y; ea[6] = e1.z; ea[7] = e1.w;
                } else {
                    const float2 c2 = Cf[ml];                 // no global load here: it would queue behind the prefetch
                    esc = c2.x; esh = c2.y;
                    const float4 x0 = *reinterpret_cast<const float4*>(&Xr[mr * F_RL + vl]);
                    const float4 x1 = *reinterpret_cast<const float4*>(&Xr[mr * F_RL + vl + 4]);
                    xa[0] = x0.x; xa[1] = x0.y; xa[2] = x0.z; xa[3] = x0.w; xa[4] = x1.x; xa[5] = x1.y; xa[6] = x1.z; xa[7] = x1.w;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) ea[jj] = 0.f;
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const bool pj = (jj < 4) ? pva : pvb;
                    const float xj = pj ? xa[jj] : 0.f;
                    float mul;                                   // statistics multiplier
                    if (EPI == FE_RESBWD) {
                        v[jj] = (pj && xj > 0.f) ? v[jj] : 0.f;
                        mul = pj ? ea[jj] : 0.f;
                    } else {
                        v[jj] = pj ? v[jj] * act_bwd(fmaf(esc, xj, esh), A.xact) : 0.f;
                        mul = xj;
                    }
                    v[jj] = stored(v[jj], DX_BF);
                    s1 += v[jj];
                    s2 = fmaf(v[jj], mul, s2);
                }
            }
            const unsigned yo = (unsigned)(mv ? ml : 0) * (unsigned)P + (unsigned)p0;
            if (mv && pva) {
                if (DX_BF) { sto2(dxs, yo, 1, v[0], v[1]); sto2(dxs, yo + 2, 1, v[2], v[3]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(dxs) + yo) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (mv && pvb) {
                if (DX_BF) { sto2(dxs, yo + 4, 1, v[4], v[5]); sto2(dxs, yo + 6, 1, v[6], v[7]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(dxs) + yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (EPI != FE_PLAIN) {
                sa[j][0] += s1; sa[j][1] += s2;
                if (flush) {
                    float t1 = sa[j][0], t2 = sa[j][1];
                    t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
                    t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
                    if (q == 0) {
                        red[((wave * U + j) * 16 + r) * 2] = mv ? t1 : 0.f;
                        red[((wave * U + j) * 16 + r) * 2 + 1] = mv ? t2 : 0.f;
                    }
                    sa[j][0] = 0.f; sa[j][1] = 0.f;
                }
            }
        }
    };

    // workgroup w owns chunks [w total / G, (w + 1) total / G); w_of(c) is the owner of chunk c.  One partial per (sample, row,
    // workgroup touching the sample): slot = workgroup - first workgroup of the sample; the sample's last workgroup clears the
    // slots nobody owns (A.tiles = ceil(G / N) + 1 slots per row, x3d_pw_bwd_fused_tiles)
    const int wg = blockIdx.x;
    auto w_of = [&](long long c) { return (int)(((c + 1) * G - 1) / total); };
    auto write_stats = [&](int n) {
        const int w_first = w_of((long long)n * cps), w_last = w_of((long long)(n + 1) * cps - 1);
        const int slot = wg - w_first;
        for (int idx = tid; idx < Ci * 2; idx += NT) {
            const int ml = idx >> 1, which = idx & 1;
            const int lt = ml >> 4, wv = (lt % MP) * 2, j = lt / MP;
            const float sv = red[((wv * U + j) * 16 + (ml & 15)) * 2 + which] +
                             red[(((wv + 1) * U + j) * 16 + (ml & 15)) * 2 + which];
            float* pp = A.partial + (((size_t)n * Ci + ml) * A.tiles) * 2 + which;
            pp[(size_t)slot * 2] = sv;
            if (wg == w_last)
                for (int t = slot + 1; t < A.tiles; ++t) pp[(size_t)t * 2] = 0.f;
        }
    };

    // Software pipeline over this workgroup's chunks: the raw rows of chunk c+1 (GEMM operands and epilogue operands)
    // are requested before the MFMAs of chunk c and converted into the LDS images after its epilogue, so every global
    // round trip has an MFMA + epilogue phase to complete; no register is live across a phase that does not need it.
    const int c_begin = (int)(((long long)wg * total) / G), c_end = (int)(((long long)(wg + 1) * total) / G);
    if (c_begin < c_end) {
        int c = c_begin;
        int n = c / cps, tile = c - n * cps;
        fetch(n, tile * F_PT);
        for (;;) {
            store(tile * F_PT);
            x3d_lds_barrier();               // chunk staged (LDS only: round 4)
            const int cn = c + 1;
            const bool more = cn < c_end;
            const int nn = more ? cn / cps : n, ntile = more ? cn - nn * cps : tile;
            if (more) fetch(nn, ntile * F_PT);
            compute_w();
            compute_d();
            const bool flush = !more || nn != n;
            epilogue(n, tile, flush);
            x3d_lds_barrier();               // everyone done reading this chunk's images; red[] complete
            if (EPI != FE_PLAIN && flush) write_stats(n);
            if (!more) break;
            c = cn; n = nn; tile = ntile;
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

// ---------------------------------------------------------------------------------------
// Forward of the stage 1-2 pointwise convs in the same architecture (round 2): persistent 8-wave workgroups over 64-voxel
// chunks, the activation rows of chunk c+1 requested before the MFMAs of chunk c, X staged as THREE bf16 planes
// (hi + mid + lo: fp32-level accuracy, six MFMA products -- pw6.hip has the argument) read transposed as the B operand, the
// pre-split forward weight pack copied into LDS once per workgroup, BN statistics per (row, chunk) in the epilogue.
// Replaces the fp32-MFMA streaming kernel pw3 (3.3-3.5 TB/s at 2 waves per SIMD) for dense inputs with K, M <= 128.
// ---------------------------------------------------------------------------------------
struct FsArgs {
    const void* x; const float* cin; int in_act;       // [N][K][P] (bf16 when x_bf); [N][K][2] or NULL
    const float* wp;                                    // forward pack: fp32 image, then bf16 hi / mid / lo planes
    void* y; float* partial;                            // [N][M][P] (bf16 when y_bf); [N][M][tiles][2] or NULL
    int N, K, M, P, tiles;
    int x_bf, y_bf;                                     // mixed-storage build only
};

template <int KP, int MP, int NWV, bool MX>
__global__ __launch_bounds__(64 * NWV, 4) void pw_fwd_stream_kernel(const FsArgs A) {
    const int x_bf = MX ? A.x_bf : 0, y_bf = MX ? A.y_bf : 0;
    constexpr int NT = 64 * NWV, RP = NT / 16;
    constexpr int NX = (KP + RP - 1) / RP;
    constexpr int KS = KP / 32;
    constexpr int MPW = NWV / 2;                           // waves per voxel half
    constexpr int U = (MP / 16 + MPW - 1) / MPW;           // units per wave: m tiles (wave >> 1) + MPW j
    __shared__ __attribute__((aligned(16))) __bf16 Xh[KP * F_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xm[KP * F_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xl[KP * F_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Wl[(MP / 16) * KS * 3 * 64 * 8];
    __shared__ float red[NWV * U * 16 * 2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int half = wave & 1, mpar = wave >> 1;
    const int P = A.P, K = A.K, M = A.M;
    const int cps = (P + F_PT - 1) / F_PT;
    const int total = A.N * cps;
    const int G = gridDim.x;
    const int c4 = (tid & 15) * 4, row0 = tid >> 4;
    const int colE = (c4 & 32) + ((c4 & 31) >> 1), colO = colE + 16;

    // forward weight pack -> LDS ([m tile][k step][plane][lane][8]), once
    {
        const int mtiles = (M + 15) / 16, kg16 = (K + 15) / 16, kg32 = (K + 31) / 32;
        const __bf16* wq = reinterpret_cast<const __bf16*>(A.wp + (size_t)mtiles * kg16 * 256);
        const size_t plane = (size_t)mtiles * kg32 * 512;
        for (int i = tid; i < (MP / 16) * KS * 3 * 64; i += NT) {
            const int ln = i & 63, rest = i >> 6;
            const int pl = rest % 3, rest2 = rest / 3;
            const int s = rest2 % KS, mt = rest2 / KS;
            const int mtc = min(mt, mtiles - 1), sc_ = min(s, kg32 - 1);      // clamped: never stored / zero X rows
            *reinterpret_cast<bf16x8*>(&Wl[(size_t)i * 8]) =
                *reinterpret_cast<const bf16x8*>(wq + pl * plane + (((size_t)mtc * kg32 + sc_) * 64 + ln) * 8);
        }
    }

    // chunk c of this workgroup's CONTIGUOUS range (see the main loop); two chunks are in flight in two register sets
    auto fetch = [&](int c, float4 (&rx)[NX], float2 (&cf)[NX]) {
        const int n = c / cps, pt = (c - n * cps) * F_PT;
        const int pc = min(pt + c4, P - 4);
        const char* xs = mx_base(A.x, (size_t)n * K * (size_t)P, x_bf);
        const float* cs = A.cin != nullptr ? A.cin + (size_t)n * K * 2 : nullptr;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const unsigned k = (unsigned)min(row0 + RP * i, K - 1);
            rx[i] = ldo4_raw(xs, k * (unsigned)P + (unsigned)pc, x_bf);      // widened in store()
            cf[i] = make_float2(1.f, 0.f);
            if (cs != nullptr) cf[i] = ldg_off<float2>(cs, k * 8u);
        }
    };
    auto store = [&](int c, const float4 (&rx)[NX], const float2 (&cf)[NX]) {
        const int pt = (c % cps) * F_PT;
        const bool pvv = pt + c4 < P;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row = row0 + RP * i;
            if (KP % RP == 0 || row < KP) {
                const bool ok = pvv && row < K;
                const float4 xw = widen4(rx[i], x_bf);
                float v[4] = {xw.x, xw.y, xw.z, xw.w};
                if (A.cin != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd(fmaf(cf[i].x, v[e], cf[i].y), A.in_act);
                }
                float xs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xs[e] = ok ? v[e] : 0.f;
                unsigned hE, mE, lE, hO, mO, lO;               // pairs (0, 2) and (1, 3): x3d_split3_pair, common.h
                x3d_split3_pair(xs[0], xs[2], hE, mE, lE);
                x3d_split3_pair(xs[1], xs[3], hO, mO, lO);
                const bf16x2 he = __builtin_bit_cast(bf16x2, hE), ho = __builtin_bit_cast(bf16x2, hO);
                const bf16x2 me = __builtin_bit_cast(bf16x2, mE), mo = __builtin_bit_cast(bf16x2, mO);
                const bf16x2 le = __builtin_bit_cast(bf16x2, lE), lo = __builtin_bit_cast(bf16x2, lO);
                *reinterpret_cast<bf16x2*>(&Xh[row * F_LD + colE]) = he;
                *reinterpret_cast<bf16x2*>(&Xh[row * F_LD + colO]) = ho;
                *reinterpret_cast<bf16x2*>(&Xm[row * F_LD + colE]) = me;
                *reinterpret_cast<bf16x2*>(&Xm[row * F_LD + colO]) = mo;
                *reinterpret_cast<bf16x2*>(&Xl[row * F_LD + colE]) = le;
                *reinterpret_cast<bf16x2*>(&Xl[row * F_LD + colO]) = lo;
            }
        }
    };

    f32x4 acc[U][2];
    const int tr_off = (8 * q + (r >> 2)) * F_LD + 4 * (r & 3) + 32 * half;
    auto tr_frag = [&](const __bf16* plane, int s, int h2) -> bf16x8 {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = plane + 32 * s * F_LD + tr_off + 16 * h2;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * F_LD));
        return cat8(v0, v1);
    };
    auto compute = [&]() {
#pragma unroll
        for (int j = 0; j < U; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) { bh[h2] = tr_frag(Xh, s, h2); bm[h2] = tr_frag(Xm, s, h2); bl[h2] = tr_frag(Xl, s, h2); }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int mt = min(mpar + MPW * j, MP / 16 - 1);
                const __bf16* wb = &Wl[((mt * KS + s) * 3 * 64 + lane) * 8];
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wb);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(wb + 64 * 8);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(wb + 2 * 64 * 8);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    // round 4: activation fragment as the A operand -> TRANSPOSED accumulator tile (pw6.hip has the argument):
                    // lane (q, r) owns channel lt * 16 + r and 8 consecutive voxels
                    acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[h2], al, acc[j][h2], 0, 0, 0);
                    acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[h2], ah, acc[j][h2], 0, 0, 0);
                    acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[h2], am, acc[j][h2], 0, 0, 0);
                    acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[h2], am, acc[j][h2], 0, 0, 0);
                    acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[h2], ah, acc[j][h2], 0, 0, 0);
                    acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[h2], ah, acc[j][h2], 0, 0, 0);
                }
            }
        }
    };
    // BN statistics of this workgroup's chunks of ONE sample, accumulated in registers: every lane sums ITS channel over ITS
    // voxels (no cross-lane step per chunk); the four lane groups of a channel are combined when the range leaves the sample
    float sa[U][2];
#pragma unroll
    for (int j = 0; j < U; ++j) { sa[j][0] = 0.f; sa[j][1] = 0.f; }
    auto epilogue = [&](int c, bool flush) {
        const int n = c / cps, tile = c - n * cps;
        const int p0 = tile * F_PT + 32 * half + 8 * q;              // 8 consecutive voxels: acc[.][jj & 1][jj >> 1]
        const bool pva = p0 < P, pvb = p0 + 4 < P;                   // P % 4 == 0: whole groups of four
        char* ys = mx_base(A.y, (size_t)n * M * (size_t)P, y_bf);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int lt = mpar + MPW * j;
            if (lt * 16 >= MP) continue;
            const int ml = lt * 16 + r;
            const bool mv = ml < M;
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = stored(((jj < 4) ? pva : pvb) ? acc[j][jj & 1][jj >> 1] : 0.f, y_bf);
            const unsigned yo = (unsigned)(mv ? ml : 0) * (unsigned)P + (unsigned)p0;
            if (mv && pva) {
                if (y_bf) { sto2(ys, yo, 1, v[0], v[1]); sto2(ys, yo + 2, 1, v[2], v[3]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(ys) + yo) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (mv && pvb) {
                if (y_bf) { sto2(ys, yo + 4, 1, v[4], v[5]); sto2(ys, yo + 6, 1, v[6], v[7]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(ys) + yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            sa[j][0] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            sa[j][1] += (fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3])) +
                        (fmaf(v[4], v[4], v[5] * v[5]) + fmaf(v[6], v[6], v[7] * v[7]));
            if (flush) {
                float s1 = sa[j][0], s2 = sa[j][1];
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (q == 0) {
                    red[((wave * U + j) * 16 + r) * 2] = mv ? s1 : 0.f;
                    red[((wave * U + j) * 16 + r) * 2 + 1] = mv ? s2 : 0.f;
                }
                sa[j][0] = 0.f; sa[j][1] = 0.f;
            }
        }
    };
    // workgroup w owns chunks [w total / G, (w + 1) total / G); w_of(c) is the owner of chunk c
    const int wg = blockIdx.x;
    auto w_of = [&](long long c) { return (int)(((c + 1) * G - 1) / total); };
    auto write_stats = [&](int n) {
        const int w_first = w_of((long long)n * cps), w_last = w_of((long long)(n + 1) * cps - 1);
        const int slot = wg - w_first;
        for (int idx = tid; idx < M * 2; idx += NT) {
            const int ml = idx >> 1, which = idx & 1;
            const int lt = ml >> 4, wv = (lt % MPW) * 2, j = lt / MPW;
            const float sv = red[((wv * U + j) * 16 + (ml & 15)) * 2 + which] +
                             red[(((wv + 1) * U + j) * 16 + (ml & 15)) * 2 + which];
            float* pp = A.partial + (((size_t)n * M + ml) * A.tiles) * 2 + which;
            pp[(size_t)slot * 2] = sv;
            if (wg == w_last)                              // the sample's last workgroup clears the slots nobody owns
                for (int t = slot + 1; t < A.tiles; ++t) pp[(size_t)t * 2] = 0.f;
        }
    };

    const int c_begin = (int)(((long long)wg * total) / G), c_end = (int)(((long long)(wg + 1) * total) / G);
    if (c_begin < c_end) {
        float4 rxa[NX], rxb[NX];
        float2 cfa[NX], cfb[NX];
        fetch(c_begin, rxa, cfa);
        if (c_begin + 1 < c_end) fetch(c_begin + 1, rxb, cfb);
        auto iter = [&](int c, float4 (&rcur)[NX], float2 (&ccur)[NX]) {
            store(c, rcur, ccur);
            x3d_lds_barrier();               // chunk staged (LDS only: round 4)
            if (c + 2 < c_end) fetch(c + 2, rcur, ccur);     // two chunks ahead, into the set just consumed
            compute();
            const bool flush = (c + 1 == c_end) || ((c + 1) / cps != c / cps);
            epilogue(c, flush && A.partial != nullptr);
            x3d_lds_barrier();               // images free; red[] complete
            if (flush && A.partial != nullptr) write_stats(c / cps);
        };
        for (int c = c_begin; c < c_end; c += 2) {
            iter(c, rxa, cfa);
            if (c + 1 < c_end) iter(c + 1, rxb, cfb);
        }
    }
}

static int fb_pad32(int c) { return (c + 31) / 32 * 32; }

// residual mode: instantiated only where its three raw tiles fit the LDS (x3d_pw_bwd_fused_ok refuses the others)
// one (epilogue, addend) pair; combinations whose LDS images do not fit are not instantiated (x3d_pw_bwd_fused_ok refuses them)
template <int CO, int CI, int EPI, int NWV, int OCC, int MX, int NS>
static void fb_launch_epi(const FbArgs& A, dim3 grid, dim3 blk, hipStream_t s) {
    if (A.addend) {
        if constexpr (fb_fits(CO, CI, EPI, true, NWV, NS))
            hipLaunchKernelGGL((pw_bwd_fused_kernel<CO, CI, EPI, true, NWV, OCC, MX, NS>), grid, blk, 0, s, A);
    } else {
        if constexpr (fb_fits(CO, CI, EPI, false, NWV, NS))
            hipLaunchKernelGGL((pw_bwd_fused_kernel<CO, CI, EPI, false, NWV, OCC, MX, NS>), grid, blk, 0, s, A);
    }
}

template <int CO, int CI, int NWV, int OCC, int MX, int NS>
static void fb_launch_res(const FbArgs& A, dim3 grid, dim3 blk, hipStream_t s) {
    if constexpr (CI <= 64) fb_launch_epi<CO, CI, FE_RESBWD, NWV, OCC, MX, NS>(A, grid, blk, s);
}

// MX = 1 (x and dx in bf16) is the conv3 of a bottleneck: activation backward, no addend
template <int CO, int CI, int NWV, int OCC, int NS>
static void fb_launch_mx1(const FbArgs& A, dim3 grid, dim3 blk, hipStream_t s) {
    if constexpr (fb_fits(CO, CI, FE_ACTBWD, false, NWV, NS))
        hipLaunchKernelGGL((pw_bwd_fused_kernel<CO, CI, FE_ACTBWD, false, NWV, OCC, 1, NS>), grid, blk, 0, s, A);
}

}  // namespace

// shapes the fused kernel is built for: dense, P % 4 == 0, padded (Co, Ci) in the instantiated set
// mx (X3D_MX_* bits): 0, X|Y (x and dx in bf16: activation-backward mode without addend only) or GA (g and a in bf16)
extern "C" int x3d_pw_bwd_fused_ok(int Cin, int Cout, int P, int mode, int has_addend, int mx) {
    if (P % 4 != 0 || P < 4 || mode < 0 || mode > 2) return 0;
    if (mode == 2 && Cin > 64) return 0;           // residual mode: three raw fp32 tiles of Cin rows must fit the LDS
    if (mx != 0 && mx != X3D_MX_GA && !(mx == (X3D_MX_X | X3D_MX_Y) && mode == 1 && !has_addend)) return 0;
    const int co = fb_pad32(Cout), ci = fb_pad32(Cin) == 96 ? 128 : fb_pad32(Cin);
    const int cop = co == 96 ? 128 : co;
    if (cop > 128 || ci > 128) return 0;
    if (cop == 32 && ci != 64) return 0;           // instantiated: (32,64) (64,32) (64,64) (64,128) (128,32) (128,64)
    if (cop == 64 && !(ci == 32 || ci == 64 || ci == 128)) return 0;
    if (cop == 128 && !(ci == 32 || ci == 64)) return 0;
    // the LDS images of this (epilogue, addend) pair must fit at the current number of operand terms (fb_fits: the
    // launcher instantiates exactly these)
    const int nwv = (cop == 128 || ci == 128) ? 16 : 8;
    const int ns = x3d_opt(X3D_OPT_BWD_TERMS) == 2 ? 2 : 3;
    if ((mode != 0 || has_addend) && Cin > fb_raw_rows(cop, ci)) return 0;       // raw epilogue tiles: see fb_raw_rows
    if (!fb_fits(cop, ci, mode, has_addend != 0, nwv, ns)) return 0;
    return 1;
}

extern "C" int x3d_pw_bwd_fused_groups(int N, int P) {
    const long long chunks = (long long)N * ((P + F_PT - 1) / F_PT);
    const int gmax = x3d_opt(X3D_OPT_FB_GRID);
    return (int)(chunks < gmax ? chunks : gmax);
}

// statistics slots per (sample, row): one per workgroup whose contiguous chunk range touches the sample (+ 1: a range may
// start inside the previous sample)
extern "C" int x3d_pw_bwd_fused_tiles(int N, int P) {
    const int G = x3d_pw_bwd_fused_groups(N, P);
    return (G + N - 1) / N + 1;
}

// mode: 0 = plain (+ addend), 1 = activation backward (x raw, xpre, xact; statistics {sum out, sum out * x}),
//       2 = residual-add + ReLU backward of the producing block (x = its output, ex = its raw conv3 output)
extern "C" int x3d_pw_bwd_fused(const void* g, const void* a, const float* cb, const float* wpacked_t, const void* x,
                                const float* xpre, int xact, int mode, const float* ex, const float* addend,
                                int addend_stride, void* dx, float* wpartial, float* partial, int N, int Cin, int Cout,
                                int T, int H, int W, int mx, void* stream) {
    X3D_CHECK_ARG(g && a && cb && wpacked_t && x && dx && wpartial);
    X3D_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(mode >= 0 && mode <= 2 && (addend_stride == 1 || addend_stride == 2));
    X3D_CHECK_ARG(mode == 0 || partial != nullptr);
    X3D_CHECK_ARG(mode != 1 || xpre != nullptr);
    X3D_CHECK_ARG(mode != 2 || (ex != nullptr && xpre == nullptr));
    const int P = T * H * W;
    if (!x3d_pw_bwd_fused_ok(Cin, Cout, P, mode, addend != nullptr, mx)) {
        x3d_set_error("x3d_pw_bwd_fused: shape Cin=%d Cout=%d P=%d mode=%d mx=%d is outside the fused kernel's set", Cin, Cout, P,
                      mode, mx);
        return X3D_EINVAL;
    }
    FbArgs A = {};
    A.g = g; A.a = a; A.cb = cb; A.x = x; A.xpre = xpre; A.xact = xact; A.wpt = wpacked_t; A.dx = dx;
    A.wpartial = wpartial; A.partial = partial; A.ex = ex; A.addend = addend; A.addend_stride = addend_stride;
    A.N = N; A.Co = Cout; A.Ci = Cin; A.P = P; A.tiles = x3d_pw_bwd_fused_tiles(N, P); A.T = T; A.H = H; A.W = W;
    A.Ho = (H - 1) / 2 + 1; A.Wo = (W - 1) / 2 + 1;
    const dim3 grid(x3d_pw_bwd_fused_groups(N, P));
    hipStream_t s = (hipStream_t)stream;
    int co = fb_pad32(Cout), ci = fb_pad32(Cin);
    if (co == 96) co = 128;
    if (ci == 96) ci = 128;
    const int ns = x3d_opt(X3D_OPT_BWD_TERMS) == 2 ? 2 : 3;
#define FB_LAUNCH2(CO_, CI_, EPI_, NWV_, OCC_, MX_, NS_) fb_launch_epi<CO_, CI_, EPI_, NWV_, OCC_, MX_, NS_>(A, grid, blk, s)
#define FB_LAUNCH3(CO_, CI_, NWV_, OCC_, MX_, NS_)                                                                       \
    do {                                                                                                                    \
        if (mode == 0) FB_LAUNCH2(CO_, CI_, FE_PLAIN, NWV_, OCC_, MX_, NS_);                                                 \
        else if (mode == 1) FB_LAUNCH2(CO_, CI_, FE_ACTBWD, NWV_, OCC_, MX_, NS_);                                           \
        else fb_launch_res<CO_, CI_, NWV_, OCC_, MX_, NS_>(A, grid, blk, s);                                                 \
    } while (0)
#define FB_LAUNCH4(CO_, CI_, NWV_, OCC_, NS_)                                                                            \
    do {                                                                                                                    \
        if (mx == 0) FB_LAUNCH3(CO_, CI_, NWV_, OCC_, 0, NS_);                                                               \
        else if (mx == X3D_MX_GA) FB_LAUNCH3(CO_, CI_, NWV_, OCC_, 2, NS_);                                                  \
        else fb_launch_mx1<CO_, CI_, NWV_, OCC_, NS_>(A, grid, blk, s);                                                      \
    } while (0)
#define FB_LAUNCH(CO_, CI_, NWV_, OCC_)                                                                                  \
    do {                                                                                                                    \
        const dim3 blk(64 * NWV_);                                                                                          \
        if (ns == 2) FB_LAUNCH4(CO_, CI_, NWV_, OCC_, 2); else FB_LAUNCH4(CO_, CI_, NWV_, OCC_, 3);                          \
    } while (0)
    x3d_note_kernel("pw_bwd_fused_kernel");
    // narrow shapes (stage 1): 4 waves, 4 workgroups per CU; wide ones (stage 2): 8 waves, 2 workgroups per CU
    if (co == 32 && ci == 64) FB_LAUNCH(32, 64, 8, 4);
    else if (co == 64 && ci == 32) FB_LAUNCH(64, 32, 8, 4);
    else if (co == 64 && ci == 64) FB_LAUNCH(64, 64, 8, 4);
    else if (co == 64 && ci == 128) FB_LAUNCH(64, 128, 16, 4);
    else if (co == 128 && ci == 32) FB_LAUNCH(128, 32, 16, 4);
    else FB_LAUNCH(128, 64, 16, 4);
#undef FB_LAUNCH
#undef FB_LAUNCH4
#undef FB_LAUNCH3
#undef FB_LAUNCH2
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

// ---- forward, stages 1-2 (pw_fwd_stream_kernel): dense, K, M <= 128, P % 4 == 0
bool x3d_pwfs_ok(int K, int M, int P) {
    const bool off = x3d_opt(X3D_OPT_NO_PWFS) != 0;
    // measured at the base shape (gpurun_out/r2/launches{19,21}.json): the expanding convs (K = 24 / 48 -> M = 54 / 108: the
    // output stream dominates) gain 20-25 % over the fp32-MFMA streaming kernel pw3; the contracting ones (K = 54 / 108 ->
    // M = 24 / 48: Swish + 3-way split of a wide input, a third of the waves idle in the MFMA phase, 100 KB of LDS at
    // K = 108) lose 15-30 % and stay on pw3
    return !off && K <= 64 && M >= K && M <= 128 && (P % 4 == 0) && P >= 4;
}

// statistics slots per sample: one per workgroup whose contiguous chunk range touches the sample
int x3d_pwfs_tiles(int N, int P) {
    const int G = x3d_pw_bwd_fused_groups(N, P);
    return (G + N - 1) / N + 1;
}

extern "C" int x3d_pw_bwd_fused_groups(int N, int P);
int x3d_pwfs_launch(const void* x, const float* cin, const float* wp, void* y, float* partial, int N, int K, int M, int P,
                    int in_act, int x_bf, int y_bf, hipStream_t s) {
    FsArgs A = {};
    A.x_bf = x_bf; A.y_bf = y_bf;
    A.x = x; A.cin = cin; A.in_act = in_act; A.wp = wp; A.y = y; A.partial = partial;
    A.N = N; A.K = K; A.M = M; A.P = P; A.tiles = x3d_pwfs_tiles(N, P);
    const dim3 grid(x3d_pw_bwd_fused_groups(N, P)), blk(512);
    const int kp = fb_pad32(K) == 96 ? 128 : fb_pad32(K), mp = fb_pad32(M) == 96 ? 128 : fb_pad32(M);
    // x3d_pwfs_ok: K <= 64 and M >= K
#define FS_GO(KP_, MP_)                                                                                      \
    do {                                                                                                     \
        if (x_bf || y_bf) hipLaunchKernelGGL((pw_fwd_stream_kernel<KP_, MP_, 8, true>), grid, blk, 0, s, A);  \
        else hipLaunchKernelGGL((pw_fwd_stream_kernel<KP_, MP_, 8, false>), grid, blk, 0, s, A);              \
    } while (0)
    x3d_note_kernel("pw_fwd_stream_kernel");
    if (kp == 32) { if (mp == 32) FS_GO(32, 32); else if (mp == 64) FS_GO(32, 64); else FS_GO(32, 128); }
    else { if (mp <= 64) FS_GO(64, 64); else FS_GO(64, 128); }
#undef FS_GO
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
