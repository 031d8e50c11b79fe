// Forward pointwise (1x1x1) convolution for the large-channel layers (stages 3-4, conv5: K >= 64, M >= 96, dense,
// P % 4 == 0) -- conv1x1x1 (x3d.py:98-103) as Bottleneck.conv1 / conv3 (:112,116,146,162) and conv5 (:231,327), with the
// producer's BN-apply + ReLU / BN*SE + Swish fused on load and the BN statistics of the output in the epilogue.
//
// These layers are small problems (10-22 MB tensors, 0.5-1 GFLOP): what bounds a kernel here is its chain of exposed
// latencies, not bytes or FLOPs.  pw4_kernel walks K in 32-channel chunks with one global round trip per chunk and does the
// GEMM on the fp32 MFMA (1/16 of the bf16 rate), 25-30 us per launch.  Here
//   * a work item is (sample, 32-voxel tile, block of <= 8 sixteen-row M tiles) and its WHOLE K is requested in one burst
//     (K x 32 voxels: 12-55 KB per workgroup, 8 waves): one global round trip per item, 1500-3000 items per launch;
//   * the activation tile is split while it is staged, x = hi + mid + lo (three bf16 = all 24 significant bits), into three
//     LDS planes in [channel][voxel] order and read TRANSPOSED (ds_read_b64_tr_b16) as the B operand; the weights come
//     pre-split the same way (x3d_pw_pack, forward image, L2 resident) through a 4-deep register ring -- the only vector
//     memory traffic during the K loop, so the in-order vmcnt never waits on anything younger;
//   * every 16x16x32 tile product is six bf16 MFMAs (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, smallest first, fp32
//     accumulate): the dropped terms are <= 2^-24 relative -- fp32 rounding level (the 2-term split of the backward
//     kernels is not accurate enough for the forward) -- at 3/8 of the fp32-MFMA cycles;
//   * wave w owns M tile w of the block and both 16-voxel column tiles: a row's statistics are complete inside one wave
//     (16-lane DPP row sum), no cross-wave combine, no second barrier.
// Voxel v of the tile sits at LDS column (v & 1) * 16 + (v >> 1), so a lane owns voxels 2r, 2r+1 (float2 stores).
#include <cstdlib>
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct P6Args {
    const void* x;        // [N][K][P] (fp32, or bf16 when x_bf)
    const float* cin;     // [N][K][2] or NULL
    const float* wp;      // forward pack: fp32 image, then bf16 hi / mid / lo planes
    void* y;              // [N][M][P] (fp32, or bf16 when y_bf)
    float* partial;       // [N][M][tiles][2] or NULL
    int N, K, M, P, tiles, in_act, mblocks, mt_run;
    int x_bf, y_bf;       // mixed-storage mode: the wide (Cmid) tensor of the conv is stored in bf16
};

constexpr int P6_BN = 32;          // voxels per item
constexpr int P6_LD = 40;          // bf16 elements per LDS row (80 B; 72 B rows + 3 workgroups per CU measured slower: the CU is throughput bound)
constexpr int P6_NT = 512;         // threads (8 waves)
constexpr int P6_RP = P6_NT / 8;   // rows staged per pass (8 lanes x float4 = one 32-voxel row)
constexpr int P6_MAXPASS = 7;      // K <= 448

__device__ __forceinline__ bf16x8 cat8_(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

#ifdef X3D_TRACE
__device__ unsigned long long g_p6trace[16384 * 8];
extern "C" int x3d_debug_p6trace(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_p6trace), bytes); }
#define P6T(i) do { if (threadIdx.x == 0) p6t[i] = wall_clock64(); } while (0)
#else
#define P6T(i) do { } while (0)
#endif

// NW: waves per workgroup -- 8, or 16 (round 4) for the K >= 320 layers of stage 4, whose three LDS planes leave room for ONE
// workgroup per CU anyway: twice the threads stage the tile (4 instead of 7 rows per thread) and every M tile of the layer
// has its own wave (12 tiles on 16 waves instead of two tiles on each of 4 of 8 waves) -- the phases of the single
// resident workgroup (staging -> barrier -> MFMA -> stores) each take half as long.
template <int IN_AFF, int NPASS, bool MX, int NW = 8>
__global__ __launch_bounds__(64 * NW, 4) void pw6_kernel(const P6Args A) {
#ifdef X3D_TRACE
    unsigned long long p6t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    P6T(0);
    const int x_bf = MX ? A.x_bf : 0, y_bf = MX ? A.y_bf : 0;      // fp32 build: folded away
    extern __shared__ __attribute__((aligned(16))) __bf16 lds6[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int K = A.K, P = A.P, M = A.M;
    const int kg32 = (K + 31) / 32, kg16 = (K + 15) / 16, Kp = kg32 * 32;
    __bf16* Xh = lds6;
    __bf16* Xm = lds6 + (size_t)Kp * P6_LD;
    __bf16* Xl = lds6 + (size_t)2 * Kp * P6_LD;

    // item = ((voxel-tile group of 8) * mblocks + mb) * 8 + (voxel tile & 7): the M blocks of one voxel tile get ids 8 apart
    // (round-robin dispatch: the same XCD, so its activation tile is fetched into that L2 once)
    const int VT = A.N * A.tiles;
    const int it = blockIdx.x;
    const int tlo = it & 7, rest = it >> 3;
    const int mb = rest % A.mblocks, vt = (rest / A.mblocks) * 8 + tlo;
    if (vt >= VT) return;
    const int n = vt / A.tiles, tile = vt - n * A.tiles;
    const int pt = tile * P6_BN;

    // ---- stage: whole K x 32 voxels in one burst
    const int c4 = (tid & 7) * 4, row0 = tid >> 3;
    constexpr int RP = 8 * NW;                 // rows staged per pass
    const int pc = min(pt + c4, P - 4);
    const bool pvv = pt + c4 < P;
    const int colE = c4 >> 1, colO = colE + 16;
    const char* xs = mx_base(A.x, (size_t)n * K * (size_t)P, x_bf);
    const float* cs = IN_AFF ? A.cin + (size_t)n * K * 2 : nullptr;
    float4 rx[NPASS];
    float2 cf[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const unsigned k = (unsigned)min(row0 + RP * i, K - 1);
        rx[i] = ldo4_raw(xs, k * (unsigned)P + (unsigned)pc, x_bf);      // widened at the staging below
        if (IN_AFF) cf[i] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cs) + k * 8u);
    }
    // this wave's A fragments: M tile (block mb, wave), planes behind the fp32 image of the pack
    const int mtiles = (M + 15) / 16;
    // (a block of more than 8 tiles -- K so large that one workgroup fills a CU's LDS, x3d_pw6_launch -- gives wave w the
    // tiles w and w + 8: the staged activation tile is then shared by all M tiles of the layer instead of being staged twice)
    int mt = min(mb * A.mt_run + wave, mtiles - 1);                            // clamped: a duplicate is never stored
    bool mt_ok = wave < A.mt_run && mb * A.mt_run + wave < mtiles;
    const __bf16* wq = reinterpret_cast<const __bf16*>(A.wp + (size_t)mtiles * kg16 * 256);
    const size_t plane = (size_t)mtiles * kg32 * 512;
    const __bf16* wa = wq + ((size_t)mt * kg32 * 64 + lane) * 8;                // + s * 512 per k step, + plane per plane
    bf16x8 ah[4], am[4], al[4];
    auto fetch_a = [&](int s, bf16x8& h, bf16x8& m, bf16x8& l) {
        const int sc = min(s, kg32 - 1);
        h = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512);
        m = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512 + plane);
        l = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512 + 2 * plane);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) fetch_a(i, ah[i], am[i], al[i]);               // in flight behind the activation burst
    P6T(1);
#ifdef X3D_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // trace build: time the round trip alone
    P6T(2);
#endif

#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int row = row0 + RP * i;
        if (row < Kp) {
            const bool ok = pvv && row < K;
            const float4 xw = widen4(rx[i], x_bf);
            float v[4] = {xw.x, xw.y, xw.z, xw.w};
            if (IN_AFF) {
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
            *reinterpret_cast<bf16x2*>(&Xh[row * P6_LD + colE]) = he;
            *reinterpret_cast<bf16x2*>(&Xh[row * P6_LD + colO]) = ho;
            *reinterpret_cast<bf16x2*>(&Xm[row * P6_LD + colE]) = me;
            *reinterpret_cast<bf16x2*>(&Xm[row * P6_LD + colO]) = mo;
            *reinterpret_cast<bf16x2*>(&Xl[row * P6_LD + colE]) = le;
            *reinterpret_cast<bf16x2*>(&Xl[row * P6_LD + colO]) = lo;
        }
    }
    P6T(3);
    __syncthreads();
    P6T(4);

    // ---- K loop: B fragments by transposed LDS reads, A fragments through the register ring
    f32x4 acc[2];
    const int tr_off = (8 * q + (r >> 2)) * P6_LD + 4 * (r & 3);
    auto tr_frag = [&](const __bf16* pl, int s, int h2) -> bf16x8 {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = pl + 32 * s * P6_LD + tr_off + 16 * h2;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * P6_LD));
        return cat8_(v0, v1);
    };
    auto step = [&](int s, const bf16x8& h, const bf16x8& m, const bf16x8& l) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const bf16x8 bh = tr_frag(Xh, s, h2), bm = tr_frag(Xm, s, h2), bl = tr_frag(Xl, s, h2);
            // round 4: the ACTIVATION fragment is the A operand (rows = voxels) and the weight fragment the B operand (the two
            // fragment layouts are the same), so the accumulator holds the TRANSPOSED tile: lane (q, r) owns channel r and
            // the voxels of rows 4 q + e -- with the interleaved LDS columns 8 consecutive voxels: two 16-byte stores per
            // M tile instead of four 8-byte ones, statistics = in-lane sums + two cross-row steps instead of 4 x 8 DPP adds
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, l, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, h, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, m, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, m, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, h, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, h, acc[h2], 0, 0, 0);
        }
    };
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {                            // second tile of this wave (blocks of more than 8 tiles): w + 8
        if (A.mt_run <= NW) break;
        mt_ok = wave + NW < A.mt_run && mb * A.mt_run + wave + NW < mtiles;
        if (!mt_ok) break;
        mt = mb * A.mt_run + wave + NW;
        wa = wq + ((size_t)mt * kg32 * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) fetch_a(i, ah[i], am[i], al[i]);
      }
      if (mt_ok) {                                // wave-uniform: waves beyond the block's tiles only helped staging
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < kg32; s0 += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (s0 + i < kg32) {
                    step(s0 + i, ah[i], am[i], al[i]);
                    if (s0 + i + 4 < kg32) fetch_a(s0 + i + 4, ah[i], am[i], al[i]);
                }
            }
        }

        P6T(5);
        // ---- epilogue: lane (q, r) holds channel mt * 16 + r and the 8 voxels pt + 8 q + j: acc[j & 1][j >> 1]
        // (voxel v of the tile sits at LDS column (v & 1) * 16 + (v >> 1); accumulator row 4 q + e of half h2 = column)
        {
            const int m = mt * 16 + r;
            const bool mv = m < M;
            const int p0 = pt + 8 * q;
            const bool pva = p0 < P, pvb = p0 + 4 < P;            // P % 4 == 0: whole groups of four
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = stored(((j < 4) ? pva : pvb) ? acc[j & 1][j >> 1] : 0.f, y_bf);
            const size_t yo = ((size_t)n * M + (mv ? m : 0)) * (size_t)P + p0;
            if (mv && pva) {
                if (y_bf) { stx2(A.y, yo, 1, v[0], v[1]); stx2(A.y, yo + 2, 1, v[2], v[3]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(A.y) + yo) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (mv && pvb) {
                if (y_bf) { stx2(A.y, yo + 4, 1, v[4], v[5]); stx2(A.y, yo + 6, 1, v[6], v[7]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(A.y) + yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (A.partial != nullptr) {
                float s1 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                float s2 = fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
                s2 += fmaf(v[4], v[4], v[5] * v[5]) + fmaf(v[6], v[6], v[7] * v[7]);
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (q == 0 && mv)
                    *reinterpret_cast<float2*>(A.partial + (((size_t)n * M + m) * A.tiles + tile) * 2) = make_float2(s1, s2);
            }
        }
      }
    }
#ifdef X3D_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        p6t[6] = wall_clock64();
        if (blockIdx.x < 16384) for (int i = 0; i < 8; ++i) g_p6trace[(size_t)blockIdx.x * 8 + i] = p6t[i];
    }
#endif
}



// ---------------------------------------------------------------------------------------
// Round 4: PERSISTENT producer / consumer form of pw6_kernel (same arithmetic, same results bit for bit).
//
// What bounded pw6: a workgroup's life is one dependent chain -- start 1.0 us, loads 0.4, staging
// (activation + three-way split: VALU) 1.6, barrier 0.6, MFMA 2.2, stores 1.0 -- and a launch is 1.5-3 rounds of two such
// workgroups per CU; the matrix pipes are busy 30 % of the time.  Here ONE workgroup per CU lives for the whole launch and
// its waves are specialised:
//   * waves 8-11 (producers) stage item i + 1: the whole K x 32 voxels of the next voxel tile requested one item ahead
//     (two register sets), BN / SE / activation applied, split hi + mid + lo, written into the OTHER of two LDS buffers;
//   * waves 0-7 (consumers) run the MFMA loop and the epilogue (statistics + stores) of item i from the first buffer:
//     wave w owns the M tiles w, w + 8 (TPW = 2: both tiles share every B fragment read from LDS); their A fragments
//     are loaded ONCE per launch and stay in registers when the layer's whole K fits (RES), else they come through a
//     register ring from the L2-resident pack as in pw6;
//   * one barrier per item: at barrier i the producers have filled buffer i & 1 and the consumers have finished reading
//     buffer (i - 1) & 1, which is the one the producers fill next.  Both roles execute exactly `niter` barriers.
// VALU staging and MFMA work of consecutive items overlap inside the workgroup, the start-up and the load latency are paid
// once per launch, and 12 waves at 3 per SIMD leave 170 registers per lane for resident A fragments.
// item = mb * VT8 + vt (voxel tile, M block): the M blocks of one voxel tile run on the same XCD (ids 8 | VT8 apart).
// ---------------------------------------------------------------------------------------
// Workgroup barrier of the item pipeline: orders the LDS buffers only.  __syncthreads() compiles to "s_waitcnt vmcnt(0)
// lgkmcnt(0); s_barrier": every barrier would drain the producers' just-issued loads of the NEXT item and the consumers'
// output stores of the previous one (measured: 3 us per item instead of 1.5).  Global memory needs no ordering here: a
// workgroup never reads what it wrote.
__device__ __forceinline__ void p8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int P8_NT = 768;          // 8 consumer + 4 producer waves
constexpr int P8_PROD0 = 512;       // first producer thread
constexpr int P8_RP = 32;           // rows staged per pass by the 256 producer threads

template <int KG, int TPW>
struct P8Ring { static constexpr int RD = (KG * TPW <= 8) ? KG : 3; };

template <int IN_AFF, int KG, int TPW>
__global__ __launch_bounds__(P8_NT, 1) void pw8_kernel(const P6Args A) {
    constexpr int RD = P8Ring<KG, TPW>::RD;               // A-fragment ring depth (k steps); RD == KG: the whole K
    extern __shared__ __attribute__((aligned(16))) __bf16 lds6[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = A.K, P = A.P, M = A.M;
    const int kg32 = (K + 31) / 32, kg16 = (K + 15) / 16, Kp = kg32 * 32;
    const size_t bufsz = (size_t)3 * Kp * P6_LD;
    const int VT = A.N * A.tiles, VT8 = (VT + 7) & ~7;
    const int items = VT8 * A.mblocks;
    const int G = (int)gridDim.x;
    const int niter = (items - (int)blockIdx.x + G - 1) / G;          // >= 1: the grid never exceeds the item count

    if (wave >= 8) {
        // ------------------------------------------------ producers
        const int pt = tid - P8_PROD0;
        const int c4 = (pt & 7) * 4, row0 = pt >> 3;
        const int colE = c4 >> 1, colO = colE + 16;
        float4 rxa[KG], rxb[KG];
        float2 cfa[KG], cfb[KG];
        bool pva = false, pvb = false;
        auto issue = [&](int i, float4 (&rx)[KG], float2 (&cf)[KG], bool& pvv) {
            const int it = (int)blockIdx.x + i * G;
            const int mb = it / VT8;
            const int vt = min(it - mb * VT8, VT - 1);
            const int n = vt / A.tiles, tile = vt - n * A.tiles;
            const int pt0 = tile * P6_BN;
            const int pc = min(pt0 + c4, P - 4);
            pvv = pt0 + c4 < P;
            const float* xs = reinterpret_cast<const float*>(A.x) + (size_t)n * K * (size_t)P;
            const float* cs = IN_AFF ? A.cin + (size_t)n * K * 2 : nullptr;
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                const unsigned k = (unsigned)min(row0 + P8_RP * j, K - 1);
                rx[j] = *reinterpret_cast<const float4*>(xs + (size_t)k * (unsigned)P + (unsigned)pc);
                if (IN_AFF) cf[j] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(cs) + k * 8u);
            }
        };
        auto stage = [&](int i, const float4 (&rx)[KG], const float2 (&cf)[KG], bool pvv) {
            __bf16* Xh = lds6 + (size_t)(i & 1) * bufsz;
            __bf16* Xm = Xh + (size_t)Kp * P6_LD;
            __bf16* Xl = Xm + (size_t)Kp * P6_LD;
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                const int row = row0 + P8_RP * j;
                if (row < Kp) {
                    const bool ok = pvv && row < K;
                    float v[4] = {rx[j].x, rx[j].y, rx[j].z, rx[j].w};
                    if (IN_AFF) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_fwd(fmaf(cf[j].x, v[e], cf[j].y), A.in_act);
                    }
                    float xs[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) xs[e] = ok ? v[e] : 0.f;
                    unsigned hE, mE, lE, hO, mO, lO;               // pairs (0, 2) and (1, 3): x3d_split3_pair, common.h
                    x3d_split3_pair(xs[0], xs[2], hE, mE, lE);
                    x3d_split3_pair(xs[1], xs[3], hO, mO, lO);
                    *reinterpret_cast<bf16x2*>(&Xh[row * P6_LD + colE]) = __builtin_bit_cast(bf16x2, hE);
                    *reinterpret_cast<bf16x2*>(&Xh[row * P6_LD + colO]) = __builtin_bit_cast(bf16x2, hO);
                    *reinterpret_cast<bf16x2*>(&Xm[row * P6_LD + colE]) = __builtin_bit_cast(bf16x2, mE);
                    *reinterpret_cast<bf16x2*>(&Xm[row * P6_LD + colO]) = __builtin_bit_cast(bf16x2, mO);
                    *reinterpret_cast<bf16x2*>(&Xl[row * P6_LD + colE]) = __builtin_bit_cast(bf16x2, lE);
                    *reinterpret_cast<bf16x2*>(&Xl[row * P6_LD + colO]) = __builtin_bit_cast(bf16x2, lO);
                }
            }
        };
        issue(0, rxa, cfa, pva);
        for (int i = 0; i < niter; i += 2) {
            if (i + 1 < niter) issue(i + 1, rxb, cfb, pvb);
            stage(i, rxa, cfa, pva);
            p8_barrier();                                         // barrier i: buffer i & 1 is full
            if (i + 1 < niter) {
                if (i + 2 < niter) issue(i + 2, rxa, cfa, pva);
                stage(i + 1, rxb, cfb, pvb);
                p8_barrier();                                     // barrier i + 1
            }
        }
        return;
    }

    // ---------------------------------------------------- consumers
    const int q = lane >> 4, r = lane & 15;
    const int mtiles = (M + 15) / 16;
    const __bf16* wq = reinterpret_cast<const __bf16*>(A.wp + (size_t)mtiles * kg16 * 256);
    const size_t plane = (size_t)mtiles * kg32 * 512;
    const int tr_off = (8 * q + (r >> 2)) * P6_LD + 4 * (r & 3);
    const bool resident = (RD == KG) && A.mblocks == 1;              // this wave's tiles never change: fetch A once
    bf16x8 ah[TPW][RD], am[TPW][RD], al[TPW][RD];
    const __bf16* wa[TPW];
    bool tok[TPW];
    int mt[TPW];
    auto set_tiles = [&](int mb) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int t = wave + 8 * j;
            tok[j] = t < A.mt_run && mb * A.mt_run + t < mtiles;
            mt[j] = min(mb * A.mt_run + t, mtiles - 1);              // clamped: a duplicate is never stored
            wa[j] = wq + ((size_t)mt[j] * kg32 * 64 + lane) * 8;     // + s * 512 per k step, + plane per plane
        }
    };
    auto fetch_a = [&](int j, int s, bf16x8& h, bf16x8& m, bf16x8& l) {
        const int sc = min(s, kg32 - 1);
        h = *reinterpret_cast<const bf16x8*>(wa[j] + (size_t)sc * 512);
        m = *reinterpret_cast<const bf16x8*>(wa[j] + (size_t)sc * 512 + plane);
        l = *reinterpret_cast<const bf16x8*>(wa[j] + (size_t)sc * 512 + 2 * plane);
    };
    auto tr_frag = [&](const __bf16* pl, int s, int h2) -> bf16x8 {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = pl + 32 * s * P6_LD + tr_off + 16 * h2;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * P6_LD));
        return cat8_(v0, v1);
    };
    int cur_mb = -1;
    for (int i = 0; i < niter; ++i) {
        const int it = (int)blockIdx.x + i * G;
        const int mb = it / VT8;
        const int vt0 = it - mb * VT8;
        const bool valid = vt0 < VT;
        const int vt = min(vt0, VT - 1);
        const int n = vt / A.tiles, tile = vt - n * A.tiles;
        const int pt0 = tile * P6_BN;
        if (!resident || i == 0) {
            if (mb != cur_mb) { set_tiles(mb); cur_mb = mb; }
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int s = 0; s < RD; ++s) fetch_a(j, s, ah[j][s], am[j][s], al[j][s]);
        }
        p8_barrier();                                             // barrier i: buffer i & 1 is full
        const __bf16* Xh = lds6 + (size_t)(i & 1) * bufsz;
        const __bf16* Xm = Xh + (size_t)Kp * P6_LD;
        const __bf16* Xl = Xm + (size_t)Kp * P6_LD;
        f32x4 acc[TPW][2];
#pragma unroll
        for (int j = 0; j < TPW; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (tok[0]) {                                                // wave-uniform (tiles ascend with j: tok[1] implies tok[0])
#pragma unroll
            for (int s = 0; s < KG; ++s) {
                if (s < kg32) {
                    constexpr int dummy = 0; (void)dummy;
                    const int slot = s % RD;
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const bf16x8 bh = tr_frag(Xh, s, h2), bm = tr_frag(Xm, s, h2), bl = tr_frag(Xl, s, h2);
#pragma unroll
                        for (int j = 0; j < TPW; ++j) {
                            if (j == 0 || tok[j]) {                  // smallest terms first, as pw6_kernel
                                // (activation fragment = A operand: transposed accumulator tile, see pw6_kernel)
                                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[j][slot], acc[j][h2], 0, 0, 0);
                                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[j][slot], acc[j][h2], 0, 0, 0);
                                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am[j][slot], acc[j][h2], 0, 0, 0);
                                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am[j][slot], acc[j][h2], 0, 0, 0);
                                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah[j][slot], acc[j][h2], 0, 0, 0);
                                acc[j][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[j][slot], acc[j][h2], 0, 0, 0);
                            }
                        }
                    }
                    if (RD < KG && s + RD < kg32) {                  // ring: the slot's next k step
#pragma unroll
                        for (int j = 0; j < TPW; ++j) fetch_a(j, s + RD, ah[j][slot], am[j][slot], al[j][slot]);
                    }
                }
            }
            // ---- epilogue: lane (q, r) holds channel mt * 16 + r and the 8 voxels pt0 + 8 q + j: acc[.][j & 1][j >> 1]
            const int p0 = pt0 + 8 * q;
            const bool pva = valid && p0 < P, pvb = valid && p0 + 4 < P;      // P % 4 == 0: whole groups of four
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (j == 0 || tok[j]) {
                    const int m = mt[j] * 16 + r;
                    const bool mv = m < M;
                    float v[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) v[jj] = ((jj < 4) ? pva : pvb) ? acc[j][jj & 1][jj >> 1] : 0.f;
                    float* yp = reinterpret_cast<float*>(A.y) + ((size_t)n * M + (mv ? m : 0)) * (size_t)P + p0;
                    if (mv && pva) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                    if (mv && pvb) *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    if (A.partial != nullptr) {
                        float s1 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                        float s2 = fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
                        s2 += fmaf(v[4], v[4], v[5] * v[5]) + fmaf(v[6], v[6], v[7] * v[7]);
                        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                        if (q == 0 && mv && valid)
                            *reinterpret_cast<float2*>(A.partial + (((size_t)n * M + m) * A.tiles + tile) * 2) = make_float2(s1, s2);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Data gradient of the same layers (ConvolutionBackward grad_input of conv1x1x1, x3d.py:98-103, fused with the BN backward
// in front and the activation / residual backward behind it exactly as pw5_kernel in pw.hip): the whole-K item structure
// above with dY = cb0*g + cb1*a + cb2 staged as TWO bf16 planes (hi + lo, 3 MFMA products: the backward kernels'
// precision), the transposed weight pack as A operand, and the epilogue operands (raw x / ReLU mask /
// statistics multiplier / addend) requested behind the first A fragments so that they land during the K loop.
// pw5_kernel walks K in 32-channel chunks with ONE global round trip per chunk (its MFMA phase is 0.1 us): 7-14 serialised
// memory latencies per item; here an item pays one.
// ---------------------------------------------------------------------------------------
enum { P7_PLAIN = 0, P7_ACTBWD = 1, P7_RESBWD = 2 };

struct P7Args {
    const void* g; const void* a; const float* cb;        // [N][K][P], [N][K][P] (bf16 when ga_bf), [N][K][3]
    const float* wp;                                       // transposed pack (M = Cin rows, K = Cout)
    void* y;                                               // [N][M][P] (bf16 when y_bf)
    float* partial;                                        // [N][M][tiles][2] (ACTBWD / RESBWD)
    const void* ex;                                        // ACTBWD: raw x (bf16 when ex_bf); RESBWD: raw conv3 output of the producing block
    int ga_bf, y_bf, ex_bf;                                // mixed-storage mode
    const float* emask;                                    // RESBWD: that block's output
    const float* ecoef; int e_act;                         // ACTBWD: [N][M][2]
    const float* addend; int addend_stride;
    int N, K, M, P, tiles, T, H, W, Ho, Wo, mblocks, mt_run;
};

// NS: bf16 terms per fp32 operand -- 3 (hi + mid + lo = all 24 significant bits, six MFMA products: fp32-level, as pw6; the
// default) or 2 (hi + lo, three products, ~2^-16 per product; option bwd_terms = 2).  The transposed pack always carries
// three planes; the two-term form reads hi and mid (mid = bf16(v - hi) is exactly its "lo").
// ADD2: the addend is the stride-(1,2,2) gradient of a downsample branch (4 launches per step): only that instantiation carries
// the (t, h, w) walk of the lane's 8 voxels -- ~18 registers the common variants, already at the 128-register cap, do not have.
template <int EPI, int NPASS, bool MX, int NS, int NW = 8, bool ADD2 = false>      // NW: waves per workgroup, see pw6_kernel
__global__ __launch_bounds__(64 * NW, (NW == 16 || NPASS <= 4) ? 4 : 2) void pw7_kernel(const P7Args A) {
    const int ga_bf = MX ? A.ga_bf : 0, y_bf = MX ? A.y_bf : 0, ex_bf = MX ? A.ex_bf : 0;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds6[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int K = A.K, P = A.P, M = A.M;
    const int kg32 = (K + 31) / 32, kg16 = (K + 15) / 16, Kp = kg32 * 32;
    __bf16* Dh = lds6;
    __bf16* Dm = lds6 + (size_t)Kp * P6_LD;                   // NS = 2: the second (last) plane
    __bf16* Dl = lds6 + (size_t)(NS - 1) * Kp * P6_LD;

    const int VT = A.N * A.tiles;
    const int it = blockIdx.x;
    const int tlo = it & 7, rest = it >> 3;
    const int mb = rest % A.mblocks, vt = (rest / A.mblocks) * 8 + tlo;
    if (vt >= VT) return;
    const int n = vt / A.tiles, tile = vt - n * A.tiles;
    const int pt = tile * P6_BN;

    // ---- stage dY: whole K x 32 voxels of g and a in one burst
    {
        const int c4 = (tid & 7) * 4, row0 = tid >> 3;
        constexpr int RP = 8 * NW;
        const int pc = min(pt + c4, P - 4);
        const bool pvv = pt + c4 < P;
        const int colE = c4 >> 1, colO = colE + 16;
        const char* gs = mx_base(A.g, (size_t)n * K * (size_t)P, ga_bf);
        const char* as = mx_base(A.a, (size_t)n * K * (size_t)P, ga_bf);
        const float* cs = A.cb + (size_t)n * K * 3;
        float4 rg[NPASS], ra[NPASS];
        float k0[NPASS], k1[NPASS], k2[NPASS];
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned k = (unsigned)min(row0 + RP * i, K - 1);
            const unsigned off = k * (unsigned)P + (unsigned)pc;
            rg[i] = ldo4_raw(gs, off, ga_bf);
            ra[i] = ldo4_raw(as, off, ga_bf);
            const float* c3 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(cs) + k * 12u);
            k0[i] = c3[0]; k1[i] = c3[1]; k2[i] = c3[2];
        }
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = row0 + RP * i;
            if (row < Kp) {
                const bool ok = pvv && row < K;
                const float4 gw = widen4(rg[i], ga_bf), aw = widen4(ra[i], ga_bf);
                const float gv[4] = {gw.x, gw.y, gw.z, gw.w}, av[4] = {aw.x, aw.y, aw.z, aw.w};
                float xs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xs[e] = ok ? fmaf(k0[i], gv[e], fmaf(k1[i], av[e], k2[i])) : 0.f;
                unsigned hE, mE, lE, hO, mO, lO;               // pairs (0, 2) and (1, 3): x3d_split3_pair, common.h
                x3d_split3_pair(xs[0], xs[2], hE, mE, lE);
                x3d_split3_pair(xs[1], xs[3], hO, mO, lO);
                const bf16x2 he = __builtin_bit_cast(bf16x2, hE), ho = __builtin_bit_cast(bf16x2, hO);
                const bf16x2 me = __builtin_bit_cast(bf16x2, mE), mo = __builtin_bit_cast(bf16x2, mO);
                const bf16x2 le = __builtin_bit_cast(bf16x2, lE), lo = __builtin_bit_cast(bf16x2, lO);
                *reinterpret_cast<bf16x2*>(&Dh[row * P6_LD + colE]) = he;
                *reinterpret_cast<bf16x2*>(&Dh[row * P6_LD + colO]) = ho;
                *reinterpret_cast<bf16x2*>(&Dm[row * P6_LD + colE]) = me;
                *reinterpret_cast<bf16x2*>(&Dm[row * P6_LD + colO]) = mo;
                if (NS == 3) {
                    *reinterpret_cast<bf16x2*>(&Dl[row * P6_LD + colE]) = le;
                    *reinterpret_cast<bf16x2*>(&Dl[row * P6_LD + colO]) = lo;
                }
            }
        }
    }

    // (the staging registers -- 8 NPASS floats of g and a -- are dead here; keep the scheduler from hoisting the A-fragment
    // and epilogue-operand requests above the staging: that is what pushes the allocation past 128 registers)
    __builtin_amdgcn_sched_barrier(0);
    // ---- this wave's A fragments (first ring entries), then the epilogue operands: both in flight across the barrier
    const int mtiles = (M + 15) / 16;
    // (blocks of more than 8 tiles -- large K, x3d_pw7_launch -- give wave w the tiles w and w + 8 of ONE staged dY tile)
    int mt = min(mb * A.mt_run + wave, mtiles - 1);                            // clamped: a duplicate is never stored
    bool mt_ok = wave < A.mt_run && mb * A.mt_run + wave < mtiles;
    const __bf16* wq = reinterpret_cast<const __bf16*>(A.wp + (size_t)mtiles * kg16 * 256);
    const size_t plane = (size_t)mtiles * kg32 * 512;
    const __bf16* wa = wq + ((size_t)mt * kg32 * 64 + lane) * 8;
    constexpr int RING = (EPI == P7_PLAIN || NPASS > 4) ? 4 : 3;   // A-fragment ring depth: three where the epilogue operands need the registers
    bf16x8 ah[RING], am[RING], al[RING];
    auto fetch_a = [&](int s, bf16x8& h, bf16x8& m, bf16x8& l) {
        const int sc = min(s, kg32 - 1);
        h = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512);
        m = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512 + plane);
        if (NS == 3) l = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512 + 2 * plane);
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) fetch_a(i, ah[i], am[i], al[i]);

    // lane (q, r) of the TRANSPOSED accumulator tile (activation fragment = A operand, see pw6_kernel): channel mt * 16 + r,
    // the 8 consecutive voxels pt + 8 q + j = acc[j & 1][j >> 1]
    const int p0 = pt + 8 * q;
    const bool pva = p0 < P, pvb = p0 + 4 < P;    // P % 4 == 0: whole groups of four
    const int pca = pva ? p0 : 0, pcb = pvb ? p0 + 4 : 0;
    const bool has_add = A.addend != nullptr;
    const bool add_s2 = ADD2 && has_add && A.addend_stride == 2;
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;
    float xv[8], mk[8], adv[8], esc1 = 1.f, esh1 = 0.f;
    auto fetch_epi = [&]() {                      // epilogue operands of tile mt: requested behind its first A fragments
        const int m = mt * 16 + r;
        const size_t mrow = (size_t)n * M + (m < M ? m : 0);
        if (EPI == P7_ACTBWD) {
            const float2 c2 = *reinterpret_cast<const float2*>(A.ecoef + mrow * 2);
            esc1 = c2.x; esh1 = c2.y;
        }
        if (EPI != P7_PLAIN) {
            if (MX && ex_bf) {                    // raw bf16 pairs, widened in the epilogue
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 t2 = ldx2_raw(A.ex, mrow * (size_t)P + ((j < 2) ? pca : pcb) + 2 * (j & 1), 1);
                    xv[2 * j] = t2.x; xv[2 * j + 1] = t2.y;
                }
            } else {
                const float* px = reinterpret_cast<const float*>(A.ex) + mrow * (size_t)P;
                const float4 ta = *reinterpret_cast<const float4*>(px + pca), tb = *reinterpret_cast<const float4*>(px + pcb);
                xv[0] = ta.x; xv[1] = ta.y; xv[2] = ta.z; xv[3] = ta.w; xv[4] = tb.x; xv[5] = tb.y; xv[6] = tb.z; xv[7] = tb.w;
            }
        }
        if (EPI == P7_RESBWD) {
            const float* pm = A.emask + mrow * (size_t)P;
            const float4 ta = *reinterpret_cast<const float4*>(pm + pca), tb = *reinterpret_cast<const float4*>(pm + pcb);
            mk[0] = ta.x; mk[1] = ta.y; mk[2] = ta.z; mk[3] = ta.w; mk[4] = tb.x; mk[5] = tb.y; mk[6] = tb.z; mk[7] = tb.w;
        }
        if (has_add) {
            const float* pa = A.addend + mrow * (size_t)addP;
            if (!ADD2 || !add_s2) {
                const float4 ta = *reinterpret_cast<const float4*>(pa + pca), tb = *reinterpret_cast<const float4*>(pa + pcb);
                adv[0] = ta.x; adv[1] = ta.y; adv[2] = ta.z; adv[3] = ta.w; adv[4] = tb.x; adv[5] = tb.y; adv[6] = tb.z; adv[7] = tb.w;
            } else {                              // stride-2 addend (downsample blocks): walk the 8 voxels' (t, h, w)
                const int hw = A.H * A.W;
                int t = pca / hw;
                const int rem = pca - t * hw;
                int h = rem / A.W, w = rem - h * A.W;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = ((j < 4) ? pva : pvb) && !(h & 1) && !(w & 1);
                    const float a = pa[ok ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : 0];
                    adv[j] = ok ? a : 0.f;
                    if (++w == A.W) { w = 0; if (++h == A.H) { h = 0; ++t; } }
                }
            }
        }
    };
    fetch_epi();
    p8_barrier();

    f32x4 acc[2];
    const int tr_off = (8 * q + (r >> 2)) * P6_LD + 4 * (r & 3);
    auto tr_frag = [&](const __bf16* pln, int s, int h2) -> bf16x8 {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = pln + 32 * s * P6_LD + tr_off + 16 * h2;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * P6_LD));
        return cat8_(v0, v1);
    };
    auto step = [&](int s, const bf16x8& h, const bf16x8& m, const bf16x8& l) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const bf16x8 bh = tr_frag(Dh, s, h2), bm = tr_frag(Dm, s, h2);
            if (NS == 3) {                        // smallest terms first, as pw6
                const bf16x8 bl = tr_frag(Dl, s, h2);
                acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, l, acc[h2], 0, 0, 0);
                acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, h, acc[h2], 0, 0, 0);
                acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, m, acc[h2], 0, 0, 0);
            }
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, m, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, h, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, h, acc[h2], 0, 0, 0);
        }
    };
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {                            // second tile of this wave (blocks of more than 8 tiles): w + 8
        if (A.mt_run <= NW) break;
        mt_ok = wave + NW < A.mt_run && mb * A.mt_run + wave + NW < mtiles;
        if (!mt_ok) break;
        mt = mb * A.mt_run + wave + NW;
        wa = wq + ((size_t)mt * kg32 * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < RING; ++i) fetch_a(i, ah[i], am[i], al[i]);
        fetch_epi();
      }
      if (mt_ok) {
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < kg32; s0 += RING) {
#pragma unroll
            for (int i = 0; i < RING; ++i) {
                if (s0 + i < kg32) {
                    step(s0 + i, ah[i], am[i], al[i]);
                    if (s0 + i + RING < kg32) fetch_a(s0 + i + RING, ah[i], am[i], al[i]);
                }
            }
        }
        {
            const int m = mt * 16 + r;
            const bool mv = m < M;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = acc[j & 1][j >> 1] + (has_add ? adv[j] : 0.f);     // (invalid voxels: masked / not stored)
            float s1 = 0.f, s2 = 0.f;
            if (EPI != P7_PLAIN) {
                float xw[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 t2 = widen2(make_float2(xv[2 * j], xv[2 * j + 1]), (MX && EPI == P7_ACTBWD) ? ex_bf : 0);
                    xw[2 * j] = t2.x; xw[2 * j + 1] = t2.y;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool pj = (j < 4) ? pva : pvb;
                    const float xj = pj ? xw[j] : 0.f;
                    if (EPI == P7_RESBWD) v[j] = (pj && mk[j] > 0.f) ? v[j] : 0.f;
                    else v[j] = pj ? v[j] * act_bwd(fmaf(esc1, xj, esh1), A.e_act) : 0.f;
                    v[j] = stored(v[j], y_bf);
                    s1 += v[j];
                    s2 = fmaf(v[j], xj, s2);
                }
            }
            const size_t yo = ((size_t)n * M + (mv ? m : 0)) * (size_t)P + p0;
            if (mv && pva) {
                if (y_bf) { stx2(A.y, yo, 1, v[0], v[1]); stx2(A.y, yo + 2, 1, v[2], v[3]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(A.y) + yo) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (mv && pvb) {
                if (y_bf) { stx2(A.y, yo + 4, 1, v[4], v[5]); stx2(A.y, yo + 6, 1, v[6], v[7]); }
                else *reinterpret_cast<float4*>(reinterpret_cast<float*>(A.y) + yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (EPI != P7_PLAIN && A.partial != nullptr) {
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (q == 0 && mv)
                    *reinterpret_cast<float2*>(A.partial + (((size_t)n * M + m) * A.tiles + tile) * 2) = make_float2(s1, s2);
            }
        }
      }
    }
}

// The ROW-wise form of the kernel above (round 3's epilogue: lane (q, r) = rows 4 q + e, voxels 2 r, 2 r + 1), kept for the
// residual-backward mode only: with three 8-voxel epilogue operands (raw a3, ReLU mask, addend) the transposed form needs
// ~150 registers and spills at the 128-register cap (measured 22.9 -> 24.5 us); this one fits with the 4-deep A ring.
template <int EPI, int NPASS, bool MX, int NS, int NW = 8>
__global__ __launch_bounds__(64 * NW, (NW == 16 || NPASS <= 4) ? 4 : 2) void pw7r_kernel(const P7Args A) {
    const int ga_bf = MX ? A.ga_bf : 0, y_bf = MX ? A.y_bf : 0, ex_bf = MX ? A.ex_bf : 0;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds6[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int K = A.K, P = A.P, M = A.M;
    const int kg32 = (K + 31) / 32, kg16 = (K + 15) / 16, Kp = kg32 * 32;
    __bf16* Dh = lds6;
    __bf16* Dm = lds6 + (size_t)Kp * P6_LD;                   // NS = 2: the second (last) plane
    __bf16* Dl = lds6 + (size_t)(NS - 1) * Kp * P6_LD;

    const int VT = A.N * A.tiles;
    const int it = blockIdx.x;
    const int tlo = it & 7, rest = it >> 3;
    const int mb = rest % A.mblocks, vt = (rest / A.mblocks) * 8 + tlo;
    if (vt >= VT) return;
    const int n = vt / A.tiles, tile = vt - n * A.tiles;
    const int pt = tile * P6_BN;

    // ---- stage dY: whole K x 32 voxels of g and a in one burst
    {
        const int c4 = (tid & 7) * 4, row0 = tid >> 3;
        constexpr int RP = 8 * NW;
        const int pc = min(pt + c4, P - 4);
        const bool pvv = pt + c4 < P;
        const int colE = c4 >> 1, colO = colE + 16;
        const char* gs = mx_base(A.g, (size_t)n * K * (size_t)P, ga_bf);
        const char* as = mx_base(A.a, (size_t)n * K * (size_t)P, ga_bf);
        const float* cs = A.cb + (size_t)n * K * 3;
        float4 rg[NPASS], ra[NPASS];
        float k0[NPASS], k1[NPASS], k2[NPASS];
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned k = (unsigned)min(row0 + RP * i, K - 1);
            const unsigned off = k * (unsigned)P + (unsigned)pc;
            rg[i] = ldo4_raw(gs, off, ga_bf);
            ra[i] = ldo4_raw(as, off, ga_bf);
            const float* c3 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(cs) + k * 12u);
            k0[i] = c3[0]; k1[i] = c3[1]; k2[i] = c3[2];
        }
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = row0 + RP * i;
            if (row < Kp) {
                const bool ok = pvv && row < K;
                const float4 gw = widen4(rg[i], ga_bf), aw = widen4(ra[i], ga_bf);
                const float gv[4] = {gw.x, gw.y, gw.z, gw.w}, av[4] = {aw.x, aw.y, aw.z, aw.w};
                float xs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xs[e] = ok ? fmaf(k0[i], gv[e], fmaf(k1[i], av[e], k2[i])) : 0.f;
                unsigned hE, mE, lE, hO, mO, lO;               // pairs (0, 2) and (1, 3): x3d_split3_pair, common.h
                x3d_split3_pair(xs[0], xs[2], hE, mE, lE);
                x3d_split3_pair(xs[1], xs[3], hO, mO, lO);
                const bf16x2 he = __builtin_bit_cast(bf16x2, hE), ho = __builtin_bit_cast(bf16x2, hO);
                const bf16x2 me = __builtin_bit_cast(bf16x2, mE), mo = __builtin_bit_cast(bf16x2, mO);
                const bf16x2 le = __builtin_bit_cast(bf16x2, lE), lo = __builtin_bit_cast(bf16x2, lO);
                *reinterpret_cast<bf16x2*>(&Dh[row * P6_LD + colE]) = he;
                *reinterpret_cast<bf16x2*>(&Dh[row * P6_LD + colO]) = ho;
                *reinterpret_cast<bf16x2*>(&Dm[row * P6_LD + colE]) = me;
                *reinterpret_cast<bf16x2*>(&Dm[row * P6_LD + colO]) = mo;
                if (NS == 3) {
                    *reinterpret_cast<bf16x2*>(&Dl[row * P6_LD + colE]) = le;
                    *reinterpret_cast<bf16x2*>(&Dl[row * P6_LD + colO]) = lo;
                }
            }
        }
    }

    // ---- this wave's A fragments (first ring entries), then the epilogue operands: both in flight across the barrier
    const int mtiles = (M + 15) / 16;
    // (blocks of more than 8 tiles -- large K, x3d_pw7_launch -- give wave w the tiles w and w + 8 of ONE staged dY tile)
    int mt = min(mb * A.mt_run + wave, mtiles - 1);                            // clamped: a duplicate is never stored
    bool mt_ok = wave < A.mt_run && mb * A.mt_run + wave < mtiles;
    const __bf16* wq = reinterpret_cast<const __bf16*>(A.wp + (size_t)mtiles * kg16 * 256);
    const size_t plane = (size_t)mtiles * kg32 * 512;
    const __bf16* wa = wq + ((size_t)mt * kg32 * 64 + lane) * 8;
    bf16x8 ah[4], am[4], al[4];
    auto fetch_a = [&](int s, bf16x8& h, bf16x8& m, bf16x8& l) {
        const int sc = min(s, kg32 - 1);
        h = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512);
        m = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512 + plane);
        if (NS == 3) l = *reinterpret_cast<const bf16x8*>(wa + (size_t)sc * 512 + 2 * plane);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) fetch_a(i, ah[i], am[i], al[i]);

    const int pl = pt + 2 * r;                    // lane (q, r): rows 4 q + e of the tile, voxels 2 r, 2 r + 1
    const bool pv = pl < P;                       // P even: both voxels or none
    const int pc2 = pv ? pl : 0;
    const bool has_add = A.addend != nullptr;
    const bool add_s2 = has_add && A.addend_stride == 2;
    int aoff[2] = {pc2, pc2 + 1};
    bool av[2] = {has_add && pv, has_add && pv};
    if (add_s2) {
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const int p = pc2 + j2;
            const int hw = A.H * A.W;
            const int t = p / hw, rem = p - t * hw;
            const int h = rem / A.W, w = rem - h * A.W;
            const bool even = !(h & 1) && !(w & 1);
            av[j2] = av[j2] && even;
            aoff[j2] = even ? (t * A.Ho + (h >> 1)) * A.Wo + (w >> 1) : 0;
        }
    }
    const long long addP = add_s2 ? (long long)A.T * A.Ho * A.Wo : (long long)P;
    float xv[4][2], mk[4][2], adv[4][2], esc[4], esh[4];
    auto fetch_epi = [&]() {                      // epilogue operands of tile mt: requested behind its first A fragments
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = mt * 16 + 4 * q + e;
        const size_t mrow = (size_t)n * M + (m < M ? m : 0);
        if (EPI == P7_ACTBWD) {
            const float2 c2 = *reinterpret_cast<const float2*>(A.ecoef + mrow * 2);
            esc[e] = c2.x; esh[e] = c2.y;
        }
        if (EPI != P7_PLAIN) {
            const float2 t2 = ldx2_raw(A.ex, mrow * (size_t)P + pc2, ex_bf);      // widened in the epilogue
            xv[e][0] = t2.x; xv[e][1] = t2.y;
        }
        if (EPI == P7_RESBWD) {
            const float2 t2 = *reinterpret_cast<const float2*>(A.emask + mrow * (size_t)P + pc2);
            mk[e][0] = t2.x; mk[e][1] = t2.y;
        }
        if (has_add) {
            const float* pa = A.addend + mrow * (size_t)addP;
            if (!add_s2) {
                const float2 t2 = *reinterpret_cast<const float2*>(pa + pc2);
                adv[e][0] = t2.x; adv[e][1] = t2.y;
            } else {
                adv[e][0] = pa[aoff[0]]; adv[e][1] = pa[aoff[1]];
            }
        }
    }
    };
    fetch_epi();
    p8_barrier();

    f32x4 acc[2];
    const int tr_off = (8 * q + (r >> 2)) * P6_LD + 4 * (r & 3);
    auto tr_frag = [&](const __bf16* pln, int s, int h2) -> bf16x8 {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* p0 = pln + 32 * s * P6_LD + tr_off + 16 * h2;
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * P6_LD));
        return cat8_(v0, v1);
    };
    auto step = [&](int s, const bf16x8& h, const bf16x8& m, const bf16x8& l) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const bf16x8 bh = tr_frag(Dh, s, h2), bm = tr_frag(Dm, s, h2);
            if (NS == 3) {                        // smallest terms first, as pw6
                const bf16x8 bl = tr_frag(Dl, s, h2);
                acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l, bh, acc[h2], 0, 0, 0);
                acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, bl, acc[h2], 0, 0, 0);
                acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m, bm, acc[h2], 0, 0, 0);
            }
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(m, bh, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, bm, acc[h2], 0, 0, 0);
            acc[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, bh, acc[h2], 0, 0, 0);
        }
    };
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {                            // second tile of this wave (blocks of more than 8 tiles): w + 8
        if (A.mt_run <= NW) break;
        mt_ok = wave + NW < A.mt_run && mb * A.mt_run + wave + NW < mtiles;
        if (!mt_ok) break;
        mt = mb * A.mt_run + wave + NW;
        wa = wq + ((size_t)mt * kg32 * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) fetch_a(i, ah[i], am[i], al[i]);
        fetch_epi();
      }
      if (mt_ok) {
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < kg32; s0 += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (s0 + i < kg32) {
                    step(s0 + i, ah[i], am[i], al[i]);
                    if (s0 + i + 4 < kg32) fetch_a(s0 + i + 4, ah[i], am[i], al[i]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = mt * 16 + 4 * q + e;
            const bool mv = m < M;
            float v[2] = {acc[0][e], acc[1][e]};
            float s1 = 0.f, s2 = 0.f;
            if (has_add) { v[0] += av[0] ? adv[e][0] : 0.f; v[1] += av[1] ? adv[e][1] : 0.f; }
            if (EPI != P7_PLAIN) {
                const float2 xw = widen2(make_float2(xv[e][0], xv[e][1]), EPI == P7_ACTBWD ? ex_bf : 0);
                const float xwv[2] = {xw.x, xw.y};
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {
                    const float xj = pv ? xwv[j2] : 0.f;
                    if (EPI == P7_RESBWD) v[j2] = (pv && mk[e][j2] > 0.f) ? v[j2] : 0.f;
                    else v[j2] = pv ? v[j2] * act_bwd(fmaf(esc[e], xj, esh[e]), A.e_act) : 0.f;
                    v[j2] = stored(v[j2], y_bf);
                    s1 += v[j2];
                    s2 = fmaf(v[j2], xj, s2);
                }
            }
            if (mv && pv) stx2(A.y, ((size_t)n * M + m) * (size_t)P + pl, y_bf, v[0], v[1]);
            if (EPI != P7_PLAIN && A.partial != nullptr) {
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if (r == 0 && mv) {
                    float* pp = A.partial + (((size_t)n * M + m) * A.tiles + tile) * 2;
                    pp[0] = s1; pp[1] = s2;
                }
            }
        }
      }
    }
}


// (Round 4 also built pw9_kernel, the persistent producer / consumer form of pw7_kernel -- bitwise the same dX, measured at
// 23-33 us against 23-26 us for pw7 at the stage 3-4 shapes and -1.5 ... -2.5 % on the step (profiles/r04/c_mb_*.txt,
// d_sweep_*.txt): the epilogue operands of the data gradient cost the registers its resident A fragments need, and a
// workgroup's phases do not overlap better than two co-resident pw7 workgroups do.  Removed again; git history has it.)

}  // namespace

// shapes pw6 takes (the caller has checked: dense, packed weights present)
bool x3d_pw6_ok(int K, int M, int P) {
    const bool off = x3d_opt(X3D_OPT_NO_PW6) != 0;
    return !off && K >= 64 && K <= P6_RP * P6_MAXPASS && M >= x3d_opt(X3D_OPT_PW6_MIN_M) && (P % 4 == 0) && P >= 4;
}

int x3d_pw6_tiles(int P) { return cdiv(P, P6_BN); }

int x3d_pw6_launch(const void* x, const float* cin, const float* wp, void* y, float* partial, int N, int K, int M,
                   int P, int in_act, int x_bf, int y_bf, hipStream_t s) {
    P6Args A = {};
    A.x = x; A.cin = cin; A.wp = wp; A.y = y; A.partial = partial; A.x_bf = x_bf; A.y_bf = y_bf;
    A.N = N; A.K = K; A.M = M; A.P = P; A.tiles = cdiv(P, P6_BN); A.in_act = in_act;
    const int mtiles = cdiv(M, 16);
    const int kp = cdiv(K, 32) * 32;
    // tiles per M block: 8 (one per wave), or 16 (two per wave) once K is so large that a workgroup's three planes take the
    // better part of a CU's LDS anyway (one resident workgroup): the layer's M tiles then share ONE staged activation tile
    const int per = kp >= x3d_opt(X3D_OPT_PW_TWO_TILES_K) ? 16 : 8;
    A.mblocks = cdiv(mtiles, per);
    A.mt_run = cdiv(mtiles, A.mblocks);
    const int VT = N * A.tiles;
    const dim3 grid(cdiv(VT, 8) * 8 * A.mblocks), block(P6_NT);
    const size_t lds = (size_t)3 * kp * P6_LD * sizeof(__bf16);
    const int npass = cdiv(kp, P6_RP);
    // more than 64 KB of dynamic LDS (K > 256) has to be allowed per kernel once
#define P6_GO(AFF, NP, MX_)                                                                                         \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw6_kernel<AFF, NP, MX_>),                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 3 * P6_RP * NP * P6_LD * 2);       \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((pw6_kernel<AFF, NP, MX_>), grid, block, lds, s, A);                                      \
    } while (0)
#define P6_PASS(AFF, MX_)                                                                                      \
    do {                                                                                                       \
        if (npass <= 2) P6_GO(AFF, 2, MX_); else if (npass <= 4) P6_GO(AFF, 4, MX_); else P6_GO(AFF, 7, MX_);  \
    } while (0)
    // round 4: the persistent producer / consumer form (fp32 storage, K <= 224: two LDS buffers of three planes each)
    if (!x_bf && !y_bf && kp <= x3d_opt(X3D_OPT_PW8_MAX_K) && !x3d_opt(X3D_OPT_NO_PW8)) {
        const int tpw = mtiles <= 8 ? 1 : 2;
        A.mblocks = cdiv(mtiles, 8 * tpw);
        A.mt_run = cdiv(mtiles, A.mblocks);
        const int vt8 = (VT + 7) & ~7;
        const int items = vt8 * A.mblocks;
        int g = x3d_opt(X3D_OPT_PW8_GRID);
        if (g <= 0) g = x3d_cu_count();
        if (g > items) g = items;
        const size_t lds8 = (size_t)2 * 3 * kp * P6_LD * sizeof(__bf16);
        const int kg = kp / 32;
#define P8_GO(AFF, KG_, TPW_)                                                                                       \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw8_kernel<AFF, KG_, TPW_>),                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * 32 * KG_ * P6_LD * 2);     \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((pw8_kernel<AFF, KG_, TPW_>), dim3(g), dim3(P8_NT), lds8, s, A);                          \
    } while (0)
#define P8_KG(AFF, TPW_)                                                                                            \
    do {                                                                                                            \
        if (kg <= 3) P8_GO(AFF, 3, TPW_); else if (kg <= 4) P8_GO(AFF, 4, TPW_);                                     \
        else if (kg <= 6) P8_GO(AFF, 6, TPW_); else P8_GO(AFF, 7, TPW_);                                             \
    } while (0)
        x3d_note_kernel("pw8_kernel");
        if (cin) { if (tpw == 1) P8_KG(1, 1); else P8_KG(1, 2); }
        else { if (tpw == 1) P8_KG(0, 1); else P8_KG(0, 2); }
#undef P8_KG
#undef P8_GO
        X3D_LAUNCH_CHECK();
        return X3D_OK;
    }
    // round 4: 16-wave workgroups (fp32 storage).  pw_waves16 = 1: the K >= 320 layers (one workgroup per CU either way);
    // = 2 (default): also every layer with more than 16 M tiles -- all M tiles of the layer (up to 32: two per wave) then
    // share ONE staged activation tile instead of one per block of 8 tiles (stage 4 conv1 / conv3 data gradient: 27 tiles,
    // four blocks: 15.0 -> 12.9 us, 19.3 -> 16.7 us); = 3: from 9 tiles on (stage 3's 14-tile layers: two blocks -> one --
    // measured SLOWER, 19.9 -> 22.0 / 25.2 -> 30.5 us: half as many workgroups per CU cost more than the second staging)
    const int w16 = x3d_opt(X3D_OPT_PW_WAVES16);
    if (!x_bf && !y_bf && kp <= 512 && mtiles <= 32 && ((w16 >= 1 && per == 16 && mtiles <= 16) || (w16 >= 2 && mtiles > 16) || (w16 >= 3 && mtiles > 8))) {
        A.mblocks = 1;
        A.mt_run = mtiles;
        const dim3 grid16(cdiv(VT, 8) * 8);
#define P6_GO16(AFF, NP)                                                                                            \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw6_kernel<AFF, NP, false, 16>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * NP * P6_LD * 2);         \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((pw6_kernel<AFF, NP, false, 16>), grid16, dim3(1024), lds, s, A);                         \
    } while (0)
#define P6_NP16(AFF) do { if (kp <= 128) P6_GO16(AFF, 1); else if (kp <= 256) P6_GO16(AFF, 2); else P6_GO16(AFF, 4); } while (0)
        x3d_note_kernel("pw6_kernel");
        if (cin) P6_NP16(1); else P6_NP16(0);
#undef P6_NP16
#undef P6_GO16
        X3D_LAUNCH_CHECK();
        return X3D_OK;
    }
    x3d_note_kernel("pw6_kernel");
    if (x_bf || y_bf) { if (cin) P6_PASS(1, true); else P6_PASS(0, true); }
    else if (cin) P6_PASS(1, false); else P6_PASS(0, false);
#undef P6_PASS
#undef P6_GO
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

// ---- data gradient (pw7_kernel): K = Cout, M = Cin
bool x3d_pw7_ok(int K, int M, int P, int mx) {
    const bool off = x3d_opt(X3D_OPT_NO_PW7) != 0;
    // measured at the base shape: K = 216 -> M = 96 (conv1 of stage 3) is the one layer where the chunked pw5_kernel is
    // not slower (23.6 vs 24.9 us: six of the eight waves own an M tile, seven k steps of staging for each)
    // (the mixed-storage mode has no chunked kernel: always here)
    // (with three-term operands, the default, pw5 is not an alternative: it is a two-term kernel)
    if (!mx && K > 128 && K < 256 && M <= 96 && x3d_opt(X3D_OPT_BWD_TERMS) == 2) return false;
    return !off && !x3d_opt(X3D_OPT_DGRAD_F32) && K >= 64 && K <= P6_RP * P6_MAXPASS && M >= 96 && (P % 4 == 0) && P >= 4;
}

int x3d_pw7_launch(const void* g, const void* a, const float* cb, const float* wpt, void* out, float* partial, int mode,
                   const void* ex, const float* emask, const float* ecoef, int e_act, const float* addend,
                   int addend_stride, int N, int K, int M, int T, int H, int W, int ga_bf, int y_bf, int ex_bf, hipStream_t s) {
    P7Args A = {};
    A.ga_bf = ga_bf; A.y_bf = y_bf; A.ex_bf = ex_bf;
    A.g = g; A.a = a; A.cb = cb; A.wp = wpt; A.y = out; A.partial = partial; A.ex = ex; A.emask = emask; A.ecoef = ecoef;
    A.e_act = e_act; A.addend = addend; A.addend_stride = addend_stride;
    A.N = N; A.K = K; A.M = M; A.P = T * H * W; A.tiles = cdiv(A.P, P6_BN); A.T = T; A.H = H; A.W = W;
    A.Ho = (H - 1) / 2 + 1; A.Wo = (W - 1) / 2 + 1;
    const int mtiles = cdiv(M, 16);
    const int kp = cdiv(K, 32) * 32;
    const int per = kp >= x3d_opt(X3D_OPT_PW_TWO_TILES_K) ? 16 : 8;      // tiles per M block (see x3d_pw6_launch)
    A.mblocks = cdiv(mtiles, per);
    A.mt_run = cdiv(mtiles, A.mblocks);
    const int VT = N * A.tiles;
    const dim3 grid(cdiv(VT, 8) * 8 * A.mblocks), block(P6_NT);
    const int ns = x3d_opt(X3D_OPT_BWD_TERMS) == 2 ? 2 : 3;
    const size_t lds = (size_t)ns * kp * P6_LD * sizeof(__bf16);
    const int npass = cdiv(kp, P6_RP);
    // ADD2 (the stride-2 addend walk): its own instantiation on the default path (fp32 storage, three terms); the mixed-storage
    // and two-term builds always carry it
    const bool add2 = addend != nullptr && addend_stride == 2;
#define P7_GO3(EPI_, NP, MX_, NS_, A2_)                                                                             \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw7_kernel<EPI_, NP, MX_, NS_, 8, A2_>),        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, NS_ * P6_RP * NP * P6_LD * 2);     \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((pw7_kernel<EPI_, NP, MX_, NS_, 8, A2_>), grid, block, lds, s, A);                        \
    } while (0)
#define P7_GOR(NP, MX_, NS_)                                                                                        \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw7r_kernel<P7_RESBWD, NP, MX_, NS_>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, NS_ * P6_RP * NP * P6_LD * 2);     \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((pw7r_kernel<P7_RESBWD, NP, MX_, NS_>), grid, block, lds, s, A);                          \
    } while (0)
#define P7_GO2(EPI_, NP, MX_, NS_)                                                                                  \
    do {                                                                                                            \
        if (EPI_ == P7_RESBWD) P7_GOR(NP, MX_, NS_);                                                                 \
        else if (MX_ || NS_ == 2) P7_GO3(EPI_, NP, MX_, NS_, true);                                                  \
        else if (add2) P7_GO3(EPI_, NP, false, 3, true);                                                             \
        else P7_GO3(EPI_, NP, false, 3, false);                                                                      \
    } while (0)
#define P7_GO(EPI_, NP, MX_) do { if (ns == 2) P7_GO2(EPI_, NP, MX_, 2); else P7_GO2(EPI_, NP, MX_, 3); } while (0)
#define P7_PASS(EPI_, MX_)                                                                                           \
    do {                                                                                                             \
        if (npass <= 2) P7_GO(EPI_, 2, MX_); else if (npass <= 4) P7_GO(EPI_, 4, MX_); else P7_GO(EPI_, 7, MX_);     \
    } while (0)
    const int w16 = x3d_opt(X3D_OPT_PW_WAVES16);                      // 16-wave workgroups: see x3d_pw6_launch
    if (!ga_bf && !y_bf && !ex_bf && ns == 3 && kp <= 512 && mtiles <= 32 &&
        ((w16 >= 1 && per == 16 && mtiles <= 16) || (w16 >= 2 && mtiles > 16) || (w16 >= 3 && mtiles > 8))) {
        A.mblocks = 1;
        A.mt_run = mtiles;
        const dim3 grid16(cdiv(VT, 8) * 8);
#define P7_GO16(EPI_, NP)                                                                                           \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw7_kernel<EPI_, NP, false, 3, 16, true>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * NP * P6_LD * 2);         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw7_kernel<EPI_, NP, false, 3, 16, false>),     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * NP * P6_LD * 2);         \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        if (addend != nullptr && addend_stride == 2)                                                                \
            hipLaunchKernelGGL((pw7_kernel<EPI_, NP, false, 3, 16, true>), grid16, dim3(1024), lds, s, A);           \
        else                                                                                                        \
            hipLaunchKernelGGL((pw7_kernel<EPI_, NP, false, 3, 16, false>), grid16, dim3(1024), lds, s, A);          \
    } while (0)
#define P7_GOR16(NP)                                                                                                \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw7r_kernel<P7_RESBWD, NP, false, 3, 16>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * NP * P6_LD * 2);         \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((pw7r_kernel<P7_RESBWD, NP, false, 3, 16>), grid16, dim3(1024), lds, s, A);               \
    } while (0)
#define P7_NP16(EPI_) do { if (kp <= 128) P7_GO16(EPI_, 1); else if (kp <= 256) P7_GO16(EPI_, 2); else P7_GO16(EPI_, 4); } while (0)
        x3d_note_kernel(mode == P7_RESBWD ? "pw7r_kernel" : "pw7_kernel");
        if (mode == P7_PLAIN) P7_NP16(P7_PLAIN);
        else if (mode == P7_ACTBWD) P7_NP16(P7_ACTBWD);
        else { if (kp <= 128) P7_GOR16(1); else if (kp <= 256) P7_GOR16(2); else P7_GOR16(4); }
#undef P7_NP16
#undef P7_GO16
#undef P7_GOR16
        X3D_LAUNCH_CHECK();
        return X3D_OK;
    }
    x3d_note_kernel(mode == P7_RESBWD ? "pw7r_kernel" : "pw7_kernel");
    if (ga_bf || y_bf || ex_bf) {
        if (mode == P7_PLAIN) P7_PASS(P7_PLAIN, true); else if (mode == P7_ACTBWD) P7_PASS(P7_ACTBWD, true); else P7_PASS(P7_RESBWD, true);
    } else {
        if (mode == P7_PLAIN) P7_PASS(P7_PLAIN, false); else if (mode == P7_ACTBWD) P7_PASS(P7_ACTBWD, false); else P7_PASS(P7_RESBWD, false);
    }
#undef P7_PASS
#undef P7_GO
#undef P7_GO2
#undef P7_GO3
#undef P7_GOR
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
