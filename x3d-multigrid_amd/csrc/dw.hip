// Channelwise 3x3x3 convolution (conv3x3x3, x3d.py:87-95; Bottleneck.conv2 x3d.py:114,150):
// forward and a fused backward (data + weight).  Every input voxel is read from HBM once and
// every output voxel written once (4.7 FLOP/B).
//
// Work decomposition: one workgroup owns CPB channels x one tile of TH rows (full width) of
// one sample and marches along T.  Two (T-plane) slots in LDS hold the activated, zero-padded
// input rows (BN-apply + ReLU is applied while staging, so padding is exact zero in the
// activated domain, as in nn.Conv3d(padding=1) on the ReLU output); every thread keeps the
// three planes its stencil touches in a register window that slides along T, so each staged
// value is read from LDS once.  Plane t+2 is requested (branch-free, clamped addresses) before
// plane t is computed and written to its slot afterwards; the step's output store is issued
// after that LDS write (no wait ever sits behind a fresh store on the single gfx9 vmcnt); one
// barrier per plane.  Each thread produces 4 consecutive outputs along W per plane from
// 9 LDS row reads; the 27 taps live in registers.
// BN statistics (sum, sum of squares) are accumulated per thread across the whole T march
// and leave as one partial per (sample, channel, tile).
// Measured (profiles/r01/timelines.txt): stage 1-2 planes run near the HBM
// roofline; on the 14x14 / 7x7 planes of stages 3-4 the kernels are bound by instruction issue
// and per-step latency, not by bytes.
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
    int tiles;                    // row tiles of a plane
    // T segments (round 3): when a launch has fewer workgroups than the chip has CUs (the 7 x 7 planes of stage 4: 216),
    // the T march is cut into `tsegs` segments of `tlen` steps, one workgroup each (one extra staged plane per later segment);
    // statistics / weight-gradient slots are indexed by blockIdx.x = seg * tiles + tile: tiles * tsegs slots per (n, c)
    int tsegs, tlen;
};

// thread-mapped grid = OUTPUT grid for forward, INPUT grid for backward
static DwGeom make_geom(int N, int C, int T, int H, int W, int stride, bool backward) {
    DwGeom g;
    g.N = N; g.C = C; g.T = T; g.H = H; g.W = W; g.stride = stride;
    g.Ho = stride == 2 ? (H - 1) / 2 + 1 : H;
    g.Wo = stride == 2 ? (W - 1) / 2 + 1 : W;
    const int GH = backward ? H : g.Ho, GW = backward ? W : g.Wo;   // thread grid
    g.groups = cdiv(GW, 4);
    const int th_max = x3d_opt(X3D_OPT_DW_TH);
    const bool balance = x3d_opt(X3D_OPT_DW_BALANCE) != 0;   // default on: +0.6 %
    int th = 256 / g.groups;
    if (th < 1) th = 1;
    if (th > th_max) th = th_max;
    if (th > GH) th = GH;
    if (balance) th = cdiv(GH, cdiv(GH, th));          // equal tile heights (same tile count)
    if (backward && stride == 2 && th > 1 && (th & 1)) th -= 1;   // even tile origin in backward stride 2
    g.TH = th;
    g.ipc = th * g.groups;
    int cpb = 256 / g.ipc;
    if (cpb < 1) cpb = 1;
    if (cpb > x3d_opt(X3D_OPT_DW_CPB_MAX)) cpb = x3d_opt(X3D_OPT_DW_CPB_MAX);
    if (cpb > C) cpb = C;
    g.cpb = cpb;
    g.tiles = cdiv(GH, th);
    const int SW = backward ? g.Wo : W;      // width of the staged tensor
    g.WP = ((SW + 3) / 4) * 4 + 8;
    if (g.WP % 32 == 0) g.WP += 4;       // rows must not alias onto the same LDS banks
    if (!backward) g.IH = (th - 1) * stride + 3;
    else g.IH = stride == 1 ? th + 2 : th / 2 + 2;
    g.slot = cpb * g.IH * g.WP;
    g.tsegs = 1; g.tlen = T;
    const long long wgs = (long long)g.tiles * cdiv(C, cpb) * N;
    // (the backward kernel's prologue + epilogue are too long for the 14 x 14 planes' 432 workgroups: 24.7 -> 29.0 us split,
    // the forward kernel's pay there: 16.9 -> 14.7 us -- separate thresholds)
    const int split_wgs = x3d_opt(backward ? X3D_OPT_DW_TSPLIT_WGS : X3D_OPT_DW_TSPLIT_WGS_FWD);
    if (T >= 8 && wgs <= (long long)split_wgs) { g.tsegs = 2; g.tlen = cdiv(T, 2); }
    // round 4: FOUR segments where two still leave fewer workgroups than `quad_wgs` (option dw_tquad_wgs[_fwd]; T >= 16)
    const int quad_wgs = x3d_opt(backward ? X3D_OPT_DW_TQUAD_WGS : X3D_OPT_DW_TQUAD_WGS_FWD);
    if (T >= 16 && 2 * wgs <= (long long)quad_wgs) { g.tsegs = 4; g.tlen = cdiv(T, 4); }
    return g;
}

struct DwFwdArgs {
    const float* x; const float* w; float* y; const float* pre; int pre_act; float* partial;
    DwGeom g;
    // training form: the producer BN's finalize is folded into this kernel (sp != NULL, pre == NULL): the scale / shift of
    // the workgroup's channels are derived from the producer conv's statistics partials sp[N][C][stiles][2]
    const float* sp; int stiles, S, count;
    const float* gamma; const float* beta; float* rmean; float* rvar; float momentum, eps;
    float* save; float* coef_out;
};

// Per-thread descriptor of one staged float4 chunk; everything that does not depend on t is
// computed once before the T march (no division in the loop).
struct Chunk {
    int goff;      // element offset of the chunk inside one (n) volume, t = 0; < 0: nothing to load
    int loff;      // float offset inside an LDS slot; < 0: chunk not owned by this thread
    int nval;      // valid elements of the float4 (0..4)
    float sc, sh;  // producer BN scale / shift of the chunk's channel
};

// Staged rows of one tile: rows row0 .. row0 + IH - 1 of a plane of SH rows; only those inside the plane are staged (the
// others stay the zeros the ring was cleared to).  ONE definition for the kernels (make_chunks) and for the host's choice of
// the per-thread chunk-slot count (nch_for): the two must agree, or `idx < total` would drop chunks silently.
__host__ __device__ inline void dw_staged_rows(int IH, int row0, int SH, int* r_lo, int* nrows) {
    const int lo = row0 < 0 ? -row0 : 0;
    const int hi = IH < SH - row0 ? IH : SH - row0;
    *r_lo = lo;
    *nrows = hi > lo ? hi - lo : 0;
}
// first staged row of a tile: forward stages the INPUT rows of its output tile, backward the OUTPUT rows of its input tile
__host__ __device__ inline int dw_tile_row0(int tile, int TH, int stride, bool backward) {
    if (!backward) return tile * TH * stride - 1;
    return stride == 1 ? tile * TH - 1 : (tile * TH) / 2 - 1;
}

template <int NCH>
__device__ __forceinline__ void make_chunks(const DwGeom& g, int n, int c0, int row0, int SH, int SW,
                                            const float* pre, Chunk (&ch)[NCH]) {
    // staged tensor: rows row0 .. row0+IH-1 of a [C][T][SH][SW] volume.  Only the rows that lie inside the tensor get a chunk:
    // a tile that covers the whole plane stages SH rows, not SH + 2 -- which is what lets the 7 x 7 planes (16 channels x
    // 9 rows x 2 chunks = 288) fit one chunk per thread.
    const int w4n = g.WP / 4 - 2;
    int r_lo, nrows;
    dw_staged_rows(g.IH, row0, SH, &r_lo, &nrows);
    const int total = g.cpb * nrows * w4n;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int idx = i * 256 + threadIdx.x;
        Chunk c;
        c.goff = -1; c.loff = -1; c.nval = 0; c.sc = 1.f; c.sh = 0.f;
        if (idx < total) {
            const int rowk = idx / w4n, w4 = idx - rowk * w4n;     // rowk = cc*nrows + k
            const int cc = rowk / nrows, ih = r_lo + rowk - cc * nrows;
            const int rowi = cc * g.IH + ih;
            const int cg = c0 + cc, hi = row0 + ih, w = w4 * 4;
            c.loff = rowi * g.WP + DW_PADL + w;
            if (cg < g.C && hi >= 0 && hi < SH && w < SW) {
                c.goff = ((cg * g.T) * SH + hi) * SW + w;
                c.nval = min(4, SW - w);
                if (pre != nullptr) { c.sc = pre[((size_t)n * g.C + cg) * 2]; c.sh = pre[((size_t)n * g.C + cg) * 2 + 1]; }
            }
        }
        ch[i] = c;
    }
}

// Branch-free: every chunk is loaded unconditionally from a clamped address (chunks outside the
// tensor / beyond T read element 0 of the sample) and masked when it is written to LDS.  A load
// under a divergent branch makes the compiler drain vmcnt -- including the previous step's output
// stores -- at the top of every T step.
// MX (mixed-storage build): `base` points at a bf16 array (the float* type is nominal); offsets are in elements.
template <int NCH, int VW, bool MX>
__device__ __forceinline__ void fetch4(const float* __restrict__ base, const Chunk (&ch)[NCH], int toff, bool tvalid,
                                       float4 (&reg)[NCH]) {
    // VW: elements per memory instruction -- 4 (rows are multiples of 4 wide: float4), 2 (even widths, e.g. the 14 x 14 planes
    // of stage 3: float2 pairs; every chunk then holds 2 or 4 valid elements and starts at an even element), 1 (odd widths)
    constexpr bool VEC = VW == 4, V2 = VW == 2;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool ok = tvalid && ch[i].goff >= 0;
        if (V2) {
            const unsigned o = ok ? (unsigned)(ch[i].goff + toff) : 0u, o2 = o + (ch[i].nval > 2 ? 2u : 0u);
            const float2 a = MX ? ldx2(base, o, 1) : *reinterpret_cast<const float2*>(base + o);
            const float2 b = MX ? ldx2(base, o2, 1) : *reinterpret_cast<const float2*>(base + o2);
            reg[i] = make_float4(a.x, a.y, b.x, b.y);
        } else if (VEC) {
            if (MX) reg[i] = ldx4(base, (size_t)(ok ? ch[i].goff + toff : 0), 1);
            else reg[i] = *reinterpret_cast<const float4*>(base + (ok ? ch[i].goff + toff : 0));
        } else if (MX) {
            const int nv = ch[i].nval;
            const unsigned o = ok ? (unsigned)(ch[i].goff + toff) : 0u;
            reg[i] = make_float4(ldx1(base, o, 1), ldx1(base, o + (nv > 1 ? 1u : 0u), 1), ldx1(base, o + (nv > 2 ? 2u : 0u), 1),
                                 ldx1(base, o + (nv > 3 ? 3u : 0u), 1));
        } else {
            // element offsets depend on the chunk only (hoisted out of the T march); a chunk that is not loaded reads
            // elements 0..3 of the sample (nval <= 4 <= the sample's size)
            const int nv = ch[i].nval;
            const unsigned o = ok ? (unsigned)(ch[i].goff + toff) : 0u;
            reg[i] = make_float4(base[o], base[o + (nv > 1 ? 1u : 0u)], base[o + (nv > 2 ? 2u : 0u)],
                                 base[o + (nv > 3 ? 3u : 0u)]);
        }
    }
}

// activation applied while storing; everything outside the tensor is exact zero
// SH (stride-1 kernels): the staged rows sit ONE column further right in LDS (data column w at DW_PADL + 1 + w), so that
// a thread's six window values w0 - 1 .. w0 + 4 start at a 16-byte boundary and come out of LDS as one b128 + one b64 --
// halo columns and zero padding included.  Round 2 fetched the two halo values from the neighbour lanes (DPP) with an LDS
// read for the wave-edge lanes and four selects per row: ~12 vector instructions per row and plane in kernels that are
// bound by vector issue.  The price is the staging write: 4 + 8 + 4 bytes instead of one b128 per chunk.
template <bool SH>
__device__ __forceinline__ void stage4(float* slot, int loff, const float4& v) {
    if (SH) {
        float* p = slot + loff + 1;
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    } else {
        *reinterpret_cast<float4*>(slot + loff) = v;
    }
}

template <int NCH, int VW, bool SH>
__device__ __forceinline__ void store_act(float* slot, const Chunk (&ch)[NCH], bool tvalid, float act_lo,
                                          const float4 (&reg)[NCH]) {
    constexpr bool VEC = VW == 4;
    // Branch-free: the activation in front of this conv is ReLU (x3d.py:147-150) or none, i.e. max(s, act_lo) with
    // act_lo = 0 / -inf (uniform); every element is computed (the registers hold real data from clamped addresses) and
    // the invalid ones are selected to the exact zero of the padding.  The stencils are VALU / issue bound: per-element
    // lane branches and the activation switch were a fifth of the instructions of a T step.
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int nv = (tvalid && ch[i].goff >= 0) ? (VEC ? 4 : ch[i].nval) : 0;
        const float sc = ch[i].sc, sh = ch[i].sh;
        float4 v;
        v.x = nv > 0 ? fmaxf(fmaf(sc, reg[i].x, sh), act_lo) : 0.f;
        v.y = nv > 1 ? fmaxf(fmaf(sc, reg[i].y, sh), act_lo) : 0.f;
        v.z = nv > 2 ? fmaxf(fmaf(sc, reg[i].z, sh), act_lo) : 0.f;
        v.w = nv > 3 ? fmaxf(fmaf(sc, reg[i].w, sh), act_lo) : 0.f;
        if (ch[i].loff >= 0) stage4<SH>(slot, ch[i].loff, v);
    }
}

// Workgroup barrier of the T march: orders the LDS ring only.  The compiler implemented the __syncthreads() at the end of a
// step as "global_store ... s_waitcnt vmcnt(0) ... s_barrier" (ISA of round 2's kernels): every step drained its own output
// store and the raw input it had just requested.  Global memory needs no ordering here: a workgroup never reads what it wrote.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float dw_act_lo(int act) { return act == X3D_ACT_RELU ? 0.f : -__builtin_inff(); }

// Forward.  LDS holds two planes (double buffer); every thread keeps the values of the three
// planes its stencil touches in registers (sliding window along T), so each staged value is
// read from LDS once per consumer instead of three times.
#ifdef X3D_TRACE
__device__ unsigned long long g_dwtrace[16384 * 8];
extern "C" int x3d_debug_dwtrace(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dwtrace), bytes); }
#define DTR(i) do { if (threadIdx.x == 0) dtr[i] = wall_clock64(); } while (0)
#else
#define DTR(i) do { } while (0)
#endif

template <int NCH, int STRIDE, bool UNI, int VW, bool MX>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DwFwdArgs A) {
    constexpr bool VEC = VW == 4, V2 = VW == 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef X3D_TRACE
    unsigned long long dtr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    DTR(0);
    constexpr int NV = (STRIDE == 1) ? 18 : 27;          // window values per plane
    const DwGeom& g = A.g;
    const int tid = threadIdx.x;
    const int slot_id = blockIdx.x;                     // seg * tiles + tile: statistics slot of this workgroup
    const int tile = slot_id % g.tiles, seg = slot_id / g.tiles, c0 = blockIdx.y * g.cpb, n = blockIdx.z;
    const int t0 = seg * g.tlen, t1 = min(g.T, t0 + g.tlen);      // this workgroup's output planes [t0, t1)
    const int ho0 = tile * g.TH;
    const int h_in0 = dw_tile_row0(tile, g.TH, STRIDE, false);
    float* ring = lds;                                  // 2 slots
    float* redbuf = lds + 2 * (size_t)g.slot;           // 256*2 floats

    for (int i = tid; i < 2 * g.slot; i += 256) ring[i] = 0.f;   // halo columns stay zero

    // UNI: one channel per workgroup (large planes) -> channel index and the 27 taps are
    // wave-uniform and live in scalar registers
    const bool active = tid < g.cpb * g.ipc;
    const int cc = (UNI || !active) ? 0 : tid / g.ipc;
    const int ri = active ? tid - cc * g.ipc : 0;
    const int row = ri / g.groups, grp = ri - row * g.groups;
    const int c = UNI ? c0 : c0 + cc;
    const bool valid = active && c < g.C && (ho0 + row) < g.Ho;

    float wt[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wt[k] = UNI ? A.w[(size_t)c0 * 27 + k] : (valid ? A.w[(size_t)c * 27 + k] : 0.f);

    Chunk ch[NCH];
    make_chunks<NCH>(g, n, c0, h_in0, g.H, g.W, A.pre, ch);
    const float* xb = reinterpret_cast<const float*>(mx_base(A.x, (size_t)n * g.C * g.T * g.H * g.W, MX));
    const int plane = g.H * g.W;

    float4 reg[NCH], reg1[NCH];
    fetch4<NCH, VW, MX>(xb, ch, t0 * plane, true, reg);              // need addresses only: in flight during the statistics below
    fetch4<NCH, VW, MX>(xb, ch, (t0 + 1) * plane, t0 + 1 < g.T, reg1);  // (two planes together: one round trip less)
    float4 regm[NCH];                                                 // a later T segment also needs plane t0 - 1
    if (t0 > 0) fetch4<NCH, VW, MX>(xb, ch, (t0 - 1) * plane, true, regm);
    if (A.sp != nullptr) {
        // BN finalize of this workgroup's channels for sample n's split (x3d.py:47-58): fp64 sums over N/S samples x
        // stiles partial pairs in a fixed order (identical in every workgroup of a (split, channel)); the tile-0
        // workgroup publishes the per-(n, c) coefficients for the backward pass, the one of the split's first sample
        // also mean / invstd and the running statistics.
        __shared__ double dred[16 * 4 * 2];
        __shared__ float lcoef[16 * 2];
        const int lane = tid & 63, wave = tid >> 6;
        const int j = n % A.S, ns = g.N / A.S, ne = ns * A.stiles;
        const int wpc = g.cpb == 1 ? 4 : (g.cpb == 2 ? 2 : 1), cpp = 4 / wpc;    // waves per channel, channels per pass
        for (int cb0 = 0; cb0 < g.cpb; cb0 += cpp) {
            const int ccs = cb0 + wave / wpc, sub = wave % wpc;
            if (ccs < g.cpb) {
                const int cg = min(c0 + ccs, g.C - 1);
                // four pairs in flight per lane (a channel's 520-784 pairs were up to twelve serial round trips); fixed order
                const int st = 64 * wpc;
                auto ldp = [&](int e) -> float2 {
                    const int ec = e < ne ? e : 0;
                    const int k = ec / A.stiles, t = ec - k * A.stiles;
                    const float2 v = *reinterpret_cast<const float2*>(A.sp + (((size_t)(j + k * A.S) * g.C + cg) * A.stiles + t) * 2);
                    return e < ne ? v : make_float2(0.f, 0.f);
                };
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
                for (int e = sub * 64 + lane; e < ne; e += 4 * st) {
                    const float2 v0 = ldp(e), v1 = ldp(e + st), v2 = ldp(e + 2 * st), v3 = ldp(e + 3 * st);
                    a0 += (double)v0.x; b0 += (double)v0.y;
                    a1 += (double)v1.x; b1 += (double)v1.y;
                    a2 += (double)v2.x; b2 += (double)v2.y;
                    a3 += (double)v3.x; b3 += (double)v3.y;
                }
                double s1 = (a0 + a1) + (a2 + a3), s2 = (b0 + b1) + (b2 + b3);
                for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (lane == 0) { dred[(ccs * 4 + sub) * 2] = s1; dred[(ccs * 4 + sub) * 2 + 1] = s2; }
            }
        }
        __syncthreads();
        if (tid < g.cpb) {
            double s1 = 0.0, s2 = 0.0;
            for (int u = 0; u < wpc; ++u) { s1 += dred[(tid * 4 + u) * 2]; s2 += dred[(tid * 4 + u) * 2 + 1]; }
            const int cg = min(c0 + tid, g.C - 1);
            const double cnt = (double)A.count * (double)ns;
            const double mean = s1 / cnt;
            double var = s2 / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            const double invstd = 1.0 / sqrt(var + (double)A.eps);
            const float scv = (float)((double)A.gamma[cg] * invstd);
            const float shv = (float)((double)A.beta[cg] - mean * (double)A.gamma[cg] * invstd);
            lcoef[tid * 2] = scv;
            lcoef[tid * 2 + 1] = shv;
            if (slot_id == 0 && c0 + tid < g.C) {
                A.coef_out[((size_t)n * g.C + cg) * 2] = scv;
                A.coef_out[((size_t)n * g.C + cg) * 2 + 1] = shv;
                if (n == j) {
                    A.save[(size_t)j * g.C + cg] = (float)mean;
                    A.save[(size_t)(A.S + j) * g.C + cg] = (float)invstd;
                    if (A.rmean != nullptr) {
                        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                        A.rmean[(size_t)j * g.C + cg] = (float)((1.0 - A.momentum) * A.rmean[(size_t)j * g.C + cg] + A.momentum * mean);
                        A.rvar[(size_t)j * g.C + cg] = (float)((1.0 - A.momentum) * A.rvar[(size_t)j * g.C + cg] + A.momentum * unb);
                    }
                }
            }
        }
        __syncthreads();
        const int chsz = g.IH * g.WP;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ci = ch[i].loff >= 0 ? ch[i].loff / chsz : 0;
            ch[i].sc = lcoef[ci * 2];
            ch[i].sh = lcoef[ci * 2 + 1];
        }
    } else {
        __syncthreads();
    }
    const float act_lo = dw_act_lo(A.pre_act);
    // ring slot of plane t: (t - t0) & 1
    store_act<NCH, VW, STRIDE == 1>(ring, ch, true, act_lo, reg);
    store_act<NCH, VW, STRIDE == 1>(ring + g.slot, ch, t0 + 1 < g.T, act_lo, reg1);
    __syncthreads();

    // LDS offset of this thread's first window element
    const int woff = cc * g.IH * g.WP + (row * STRIDE) * g.WP + DW_PADL + grp * 4 * STRIDE - 1;
    // stride 1: the halo columns come from the neighbour lanes (same row: consecutive threads); only the first / last lane
    // of a wave in the middle of a row still reads its one missing element from LDS -- one read per row whose other lanes
    // fetch a conflict-free dummy word; the first / last group of a row takes the zero padding
    auto read_plane = [&](const float* slot, float (&v)[NV]) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float* rp = slot + woff + kh * g.WP;
            if (STRIDE == 1) {
                // shifted rows (stage4): window column w0 - 1 sits at the aligned word woff + 1
                const float4 a = *reinterpret_cast<const float4*>(rp + 1);
                const float2 b = *reinterpret_cast<const float2*>(rp + 5);
                v[kh * 6] = a.x; v[kh * 6 + 1] = a.y; v[kh * 6 + 2] = a.z; v[kh * 6 + 3] = a.w;
                v[kh * 6 + 4] = b.x; v[kh * 6 + 5] = b.y;
            } else {
                v[kh * 9] = rp[0];
                const float4 a = *reinterpret_cast<const float4*>(rp + 1);
                const float4 b = *reinterpret_cast<const float4*>(rp + 5);
                v[kh * 9 + 1] = a.x; v[kh * 9 + 2] = a.y; v[kh * 9 + 3] = a.z; v[kh * 9 + 4] = a.w;
                v[kh * 9 + 5] = b.x; v[kh * 9 + 6] = b.y; v[kh * 9 + 7] = b.z; v[kh * 9 + 8] = b.w;
            }
        }
    };

    float s1 = 0.f, s2 = 0.f;
    const size_t ybase = (((size_t)n * g.C + c) * g.T) * (size_t)g.Ho * g.Wo + (size_t)(ho0 + row) * g.Wo + grp * 4;

    // one T step: window planes (wa, wb, wc) = (t-1, t, t+1).
    // Order matters on gfx9-family hardware (ONE vmcnt for loads and stores, and the compiler waits
    // vmcnt(0) for a load whenever a store may also be pending): the plane t+2 loads are consumed
    // (written to LDS) BEFORE this step's output store is issued, so no wait ever sits behind a
    // fresh store; the store's ack then hides under the next step's stencil.
    auto step = [&](int t, float (&wa)[NV], float (&wb)[NV], float (&wc)[NV]) {
        if (t == 5) DTR(2);
        const bool more = t + 2 < g.T && t + 2 <= t1;                  // plane t + 2 feeds output t + 1 < t1
        fetch4<NCH, VW, MX>(xb, ch, (t + 2) * plane, more, reg);       // in flight during the stencil
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            read_plane(ring + (size_t)((t + 1 - t0) & 1) * g.slot, wc);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const float(&v)[NV] = kt == 0 ? wa : (kt == 1 ? wb : wc);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float w0 = wt[kt * 9 + kh * 3], w1 = wt[kt * 9 + kh * 3 + 1], w2 = wt[kt * 9 + kh * 3 + 2];
                    constexpr int RS = (STRIDE == 1) ? 6 : 9;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        o[i] = fmaf(w0, v[kh * RS + i * STRIDE], fmaf(w1, v[kh * RS + i * STRIDE + 1],
                                    fmaf(w2, v[kh * RS + i * STRIDE + 2], o[i])));
                }
            }
            if (!VEC) {
                const int wo = grp * 4;
#pragma unroll
                for (int i = 1; i < 4; ++i) if (wo + i >= g.Wo) o[i] = 0.f;
            }
            if (MX) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = stored(o[i], 1);      // statistics of the tensor as stored
            }
            s1 += (o[0] + o[1]) + (o[2] + o[3]);
            s2 = fmaf(o[0], o[0], fmaf(o[1], o[1], fmaf(o[2], o[2], fmaf(o[3], o[3], s2))));
        }
        if (t == 5) DTR(3);
        // the slot of plane t, last read one barrier ago -> free for plane t+2
        store_act<NCH, VW, STRIDE == 1>(ring + (size_t)((t - t0) & 1) * g.slot, ch, more, act_lo, reg);
        if (t == 5) DTR(4);
        if (valid) {
            float* py = A.y + ybase + (size_t)t * g.Ho * g.Wo;
            const int wo = grp * 4;
            if (MX) {
                const size_t yi = ybase + (size_t)t * g.Ho * g.Wo;
                if (VEC) {
                    stx4(A.y, yi, 1, o[0], o[1], o[2], o[3]);
                } else if (V2) {
                    stx2(A.y, yi, 1, o[0], o[1]);
                    if (wo + 2 < g.Wo) stx2(A.y, yi + 2, 1, o[2], o[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (wo + i < g.Wo) stx1(A.y, yi + i, 1, o[i]);
                }
            } else if (VEC) {
                *reinterpret_cast<float4*>(py) = make_float4(o[0], o[1], o[2], o[3]);
            } else if (V2) {
                *reinterpret_cast<float2*>(py) = make_float2(o[0], o[1]);
                if (wo + 2 < g.Wo) *reinterpret_cast<float2*>(py + 2) = make_float2(o[2], o[3]);
            } else {
                py[0] = o[0];
                if (wo + 1 < g.Wo) py[1] = o[1];
                if (wo + 2 < g.Wo) py[2] = o[2];
                if (wo + 3 < g.Wo) py[3] = o[3];
            }
        }
        lds_barrier();
        if (t == 5) DTR(5);
    };

    float w0[NV], w1[NV], w2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { w0[i] = 0.f; w1[i] = 0.f; w2[i] = 0.f; }
    if (valid) read_plane(ring, w1);                     // plane t0
    __syncthreads();       // the first step overwrites slot 0: every wave must have read plane t0 first
    if (t0 > 0) {
        // a later T segment: plane t0 - 1 is part of the first output's window -- staged through the free slot 0 (plane
        // t0 + 1 sits in slot 1), read into the window, then slot 0 goes back to the march
        store_act<NCH, VW, STRIDE == 1>(ring, ch, true, act_lo, regm);
        __syncthreads();
        if (valid) read_plane(ring, w0);
        __syncthreads();
    }
    DTR(1);
    for (int t = t0; t < t1; t += 3) {
        step(t, w0, w1, w2);
        if (t + 1 < t1) step(t + 1, w1, w2, w0);
        if (t + 2 < t1) step(t + 2, w2, w0, w1);
    }
    DTR(6);

    if (A.partial != nullptr) {
        redbuf[tid * 2] = valid ? s1 : 0.f;
        redbuf[tid * 2 + 1] = valid ? s2 : 0.f;
        __syncthreads();
        if (tid < g.cpb * 2) {
            const int chn = tid >> 1, which = tid & 1;
            if (c0 + chn < g.C) {
                float s = 0.f;
                for (int i = 0; i < g.ipc; ++i) s += redbuf[(chn * g.ipc + i) * 2 + which];
                A.partial[(((size_t)n * g.C + c0 + chn) * (g.tiles * g.tsegs) + slot_id) * 2 + which] = s;
            }
        }
    }
#ifdef X3D_TRACE
    if (tid == 0) {
        dtr[7] = wall_clock64();
        const size_t id = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (id < 16384) for (int i = 0; i < 8; ++i) g_dwtrace[id * 8 + i] = dtr[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------
// Fused backward.  Thread grid = INPUT voxels (4 consecutive w per thread); the LDS planes hold
// dY = cb0*g + cb1*a + cb2 at OUTPUT resolution, zero outside the tensor.  Same double buffer +
// register sliding window as the forward.
// ---------------------------------------------------------------------------------------
struct DwBwdArgs {
    const float* g; const float* a; const float* cb; const float* w;
    const float* x; const float* pre; int pre_act;
    float* out; float* wpartial; float* partial;
    DwGeom geo;
    // cb == NULL: the producer BN's backward finalize (one split) is folded into this kernel: A, B, C of the workgroup's
    // channels are derived from the statistics partials sp[N][C][stiles][2] = {sum g, sum g*a} of the kernel that wrote g
    const float* sp; int stiles, count;
    const float* gamma; const float* save; float* dgamma; float* dbeta;
};

template <int NCH, int VW, bool SH>
__device__ __forceinline__ void store_dy(float* slot, const Chunk (&ch)[NCH], bool tvalid, const float (&k0)[NCH],
                                         const float (&k1)[NCH], const float (&k2)[NCH], const float4 (&rg)[NCH],
                                         const float4 (&ra)[NCH]) {
    constexpr bool VEC = VW == 4;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (ch[i].loff >= 0) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tvalid && ch[i].goff >= 0) {
                v.x = fmaf(k0[i], rg[i].x, fmaf(k1[i], ra[i].x, k2[i]));
                if (VEC || ch[i].nval > 1) v.y = fmaf(k0[i], rg[i].y, fmaf(k1[i], ra[i].y, k2[i]));
                if (VEC || ch[i].nval > 2) v.z = fmaf(k0[i], rg[i].z, fmaf(k1[i], ra[i].z, k2[i]));
                if (VEC || ch[i].nval > 3) v.w = fmaf(k0[i], rg[i].w, fmaf(k1[i], ra[i].w, k2[i]));
            }
            stage4<SH>(slot, ch[i].loff, v);
        }
    }
}

template <int NCH, int STRIDE, bool UNI, int VW, bool MX>
__global__ __launch_bounds__(256) void dw_bwd_kernel(const DwBwdArgs A) {
    constexpr bool VEC = VW == 4, V2 = VW == 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef X3D_TRACE
    unsigned long long dtr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    DTR(0);
    constexpr int NV = (STRIDE == 1) ? 18 : 6;           // window values per plane
    const DwGeom& g = A.geo;
    const int tid = threadIdx.x;
    const int slot_id = blockIdx.x;                               // seg * tiles + tile
    const int tile = slot_id % g.tiles, seg = slot_id / g.tiles, c0 = blockIdx.y * g.cpb, n = blockIdx.z;
    const int t0 = seg * g.tlen, t1 = min(g.T, t0 + g.tlen);      // this workgroup's planes [t0, t1)
    const int h0 = tile * g.TH;                                   // first input row of the tile
    const int ho_lo = dw_tile_row0(tile, g.TH, STRIDE, true);     // first staged output row
    float* ring = lds;

    for (int i = tid; i < 2 * g.slot; i += 256) ring[i] = 0.f;

    const bool active = tid < g.cpb * g.ipc;
    const int cc = (UNI || !active) ? 0 : tid / g.ipc;
    const int ri = active ? tid - cc * g.ipc : 0;
    const int row = ri / g.groups, grp = ri - row * g.groups;
    const int c = UNI ? c0 : c0 + cc, h = h0 + row, w0 = grp * 4;
    const bool valid = active && c < g.C && h < g.H;

    float wt[27], dwacc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        wt[k] = UNI ? A.w[(size_t)c0 * 27 + k] : (valid ? A.w[(size_t)c * 27 + k] : 0.f);
        dwacc[k] = 0.f;
    }
    const float act_lo = dw_act_lo(A.pre_act);
    float sc = 1.f, sh = 0.f;
    if (A.pre != nullptr && (UNI || valid)) { sc = A.pre[((size_t)n * g.C + c) * 2]; sh = A.pre[((size_t)n * g.C + c) * 2 + 1]; }

    Chunk ch[NCH];
    make_chunks<NCH>(g, n, c0, ho_lo, g.Ho, g.Wo, nullptr, ch);
    float k0[NCH], k1[NCH], k2[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        k0[i] = k1[i] = k2[i] = 0.f;
        if (A.cb != nullptr && ch[i].goff >= 0) {
            const int cg = ch[i].goff / (g.T * g.Ho * g.Wo);
            const float* cb = A.cb + ((size_t)n * g.C + cg) * 3;
            k0[i] = cb[0]; k1[i] = cb[1]; k2[i] = cb[2];
        }
    }
    const size_t vol_o = (size_t)n * g.C * g.T * g.Ho * g.Wo;
    const float* gb = reinterpret_cast<const float*>(mx_base(A.g, vol_o, MX));
    const float* ab = reinterpret_cast<const float*>(mx_base(A.a, vol_o, MX));
    const int plane_o = g.Ho * g.Wo;

    float4 rg[NCH], ra[NCH];
    fetch4<NCH, VW, MX>(gb, ch, t0 * plane_o, true, rg);           // addresses only: in flight during the statistics below
    fetch4<NCH, VW, MX>(ab, ch, t0 * plane_o, true, ra);
    float4 rgm[NCH], ram[NCH];                                     // a later T segment also needs dY plane t0 - 1
    if (t0 > 0) {
        fetch4<NCH, VW, MX>(gb, ch, (t0 - 1) * plane_o, true, rgm);
        fetch4<NCH, VW, MX>(ab, ch, (t0 - 1) * plane_o, true, ram);
    }
    if (A.cb == nullptr) {
        // BN backward finalize of this workgroup's channels (single split; x3d.py:47-58 backward): fp64 sums over all
        // samples x stiles partial pairs in a fixed order (identical in every workgroup of a channel); the (tile 0,
        // sample 0) workgroup writes dgamma / dbeta.
        __shared__ double dred[16 * 4 * 2];
        __shared__ float lcb[16 * 3];
        const int lane = tid & 63, wave = tid >> 6;
        const int ne = g.N * A.stiles;
        const int wpc = g.cpb == 1 ? 4 : (g.cpb == 2 ? 2 : 1), cpp = 4 / wpc;    // waves per channel, channels per pass
        for (int cb0 = 0; cb0 < g.cpb; cb0 += cpp) {
            const int ccs = cb0 + wave / wpc, sub = wave % wpc;
            if (ccs < g.cpb) {
                const int cg = min(c0 + ccs, g.C - 1);
                const int st = 64 * wpc;                       // four pairs in flight per lane (see dw_fwd_kernel)
                auto ldp = [&](int e) -> float2 {
                    const int ec = e < ne ? e : 0;
                    const int k = ec / A.stiles, t = ec - k * A.stiles;
                    const float2 v = *reinterpret_cast<const float2*>(A.sp + (((size_t)k * g.C + cg) * A.stiles + t) * 2);
                    return e < ne ? v : make_float2(0.f, 0.f);
                };
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
                for (int e = sub * 64 + lane; e < ne; e += 4 * st) {
                    const float2 v0 = ldp(e), v1 = ldp(e + st), v2 = ldp(e + 2 * st), v3 = ldp(e + 3 * st);
                    a0 += (double)v0.x; b0 += (double)v0.y;
                    a1 += (double)v1.x; b1 += (double)v1.y;
                    a2 += (double)v2.x; b2 += (double)v2.y;
                    a3 += (double)v3.x; b3 += (double)v3.y;
                }
                double s1 = (a0 + a1) + (a2 + a3), s2 = (b0 + b1) + (b2 + b3);
                for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
                if (lane == 0) { dred[(ccs * 4 + sub) * 2] = s1; dred[(ccs * 4 + sub) * 2 + 1] = s2; }
            }
        }
        __syncthreads();
        if (tid < g.cpb) {
            double sg = 0.0, sga = 0.0;
            for (int u = 0; u < wpc; ++u) { sg += dred[(tid * 4 + u) * 2]; sga += dred[(tid * 4 + u) * 2 + 1]; }
            const int cg = min(c0 + tid, g.C - 1);
            const double M = (double)A.count * (double)g.N;
            const double mean = A.save[cg], invstd = A.save[(size_t)g.C + cg];
            const double sgx = (sga - mean * sg) * invstd;
            const double k = (double)A.gamma[cg] * invstd;
            lcb[tid * 3] = (float)k;
            lcb[tid * 3 + 1] = (float)(-k * invstd * sgx / M);
            lcb[tid * 3 + 2] = (float)(-k * sg / M + k * invstd * mean * sgx / M);
            if (slot_id == 0 && n == 0 && c0 + tid < g.C) {
                A.dgamma[cg] = (float)sgx;
                A.dbeta[cg] = (float)sg;
            }
        }
        __syncthreads();                                   // also orders the ring zero-fill before the staging below
        const int chvol = g.T * g.Ho * g.Wo;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (ch[i].goff >= 0) {
                const int ci = ch[i].goff / chvol - c0;
                k0[i] = lcb[ci * 3]; k1[i] = lcb[ci * 3 + 1]; k2[i] = lcb[ci * 3 + 2];
            }
        }
    } else {
        __syncthreads();
    }
    {
        float4 rg1[NCH], ra1[NCH];                       // the second plane requested before the first is staged: one round trip less
        fetch4<NCH, VW, MX>(gb, ch, (t0 + 1) * plane_o, t0 + 1 < g.T, rg1);
        fetch4<NCH, VW, MX>(ab, ch, (t0 + 1) * plane_o, t0 + 1 < g.T, ra1);
        store_dy<NCH, VW, STRIDE == 1>(ring, ch, true, k0, k1, k2, rg, ra);
        store_dy<NCH, VW, STRIDE == 1>(ring + g.slot, ch, t0 + 1 < g.T, k0, k1, k2, rg1, ra1);
    }
    __syncthreads();

    // window layout.  stride 1: rows ho = h+1-kh (kh = 0,1,2) -> staged rows row+2-kh, 6 columns
    // starting at wo = w0-1.  stride 2: rows with (h+1-kh) even: h even -> kh=1 (one row, slot 0 of the
    // window), h odd -> kh=0 (slot 0) and kh=2 (slot 1); 3 columns wo = 2grp .. 2grp+2.
    const int par = h & 1;
    int roff0, roff1;                                    // LDS offsets of the window rows
    if (STRIDE == 1) {
        roff0 = cc * g.IH * g.WP + (row + 2) * g.WP + DW_PADL + w0 - 1;     // kh = 0; kh adds -WP
        roff1 = 0;
    } else {
        const int hoA = par ? (h + 1) >> 1 : h >> 1;     // kh = 0 (odd h) or kh = 1 (even h)
        const int hoB = (h - 1) >> 1;                    // kh = 2 (odd h only)
        roff0 = cc * g.IH * g.WP + (hoA - ho_lo) * g.WP + DW_PADL + 2 * grp;
        roff1 = cc * g.IH * g.WP + ((par ? hoB : hoA) - ho_lo) * g.WP + DW_PADL + 2 * grp;
    }
    auto read_plane = [&](const float* slot, float (&v)[NV]) {
        if (STRIDE == 1) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* rp = slot + roff0 - kh * g.WP;          // shifted rows (stage4): column w0 - 1 at the aligned word
                const float4 a = *reinterpret_cast<const float4*>(rp + 1);
                const float2 b = *reinterpret_cast<const float2*>(rp + 5);
                v[kh * 6] = a.x; v[kh * 6 + 1] = a.y; v[kh * 6 + 2] = a.z; v[kh * 6 + 3] = a.w;
                v[kh * 6 + 4] = b.x; v[kh * 6 + 5] = b.y;
            }
        } else {
            const float* ra_ = slot + roff0;
            const float* rb_ = slot + roff1;
            v[0] = ra_[0]; v[1] = ra_[1]; v[2] = ra_[2];
            v[3] = rb_[0]; v[4] = rb_[1]; v[5] = rb_[2];
        }
    };

    float s1 = 0.f, s2 = 0.f;
    const size_t xbase = (((size_t)n * g.C + c) * g.T) * (size_t)g.H * g.W + (size_t)h * g.W + w0;

    // raw input of one step: unconditional load from a clamped address (see fetch4), one step ahead
    float xnext[4];
    auto load_x = [&](int t) {
        const bool ok = valid && t < g.T;
        const float* px = A.x + (ok ? xbase + (size_t)t * g.H * g.W : 0);
        if (MX) {
            const size_t xi = ok ? xbase + (size_t)t * g.H * g.W : 0;
            if (VEC) {
                const float4 qv = ldx4(A.x, xi, 1);
                xnext[0] = qv.x; xnext[1] = qv.y; xnext[2] = qv.z; xnext[3] = qv.w;
            } else if (V2) {
                const int rem = g.W - w0;
                const float2 qa = ldx2(A.x, xi, 1), qb = ldx2(A.x, xi + (rem > 2 ? 2 : 0), 1);
                xnext[0] = qa.x; xnext[1] = qa.y; xnext[2] = rem > 2 ? qb.x : 0.f; xnext[3] = rem > 2 ? qb.y : 0.f;
            } else {
                const int rem = g.W - w0;
                xnext[0] = ldx1(A.x, xi, 1); xnext[1] = ldx1(A.x, xi + (rem > 1 ? 1 : 0), 1);
                xnext[2] = ldx1(A.x, xi + (rem > 2 ? 2 : 0), 1); xnext[3] = ldx1(A.x, xi + (rem > 3 ? 3 : 0), 1);
#pragma unroll
                for (int i = 1; i < 4; ++i) xnext[i] = i < rem ? xnext[i] : 0.f;
            }
        } else if (VEC) {
            const float4 qv = *reinterpret_cast<const float4*>(px);
            xnext[0] = qv.x; xnext[1] = qv.y; xnext[2] = qv.z; xnext[3] = qv.w;
        } else if (V2) {
            const int rem = g.W - w0;          // even; a thread beyond the row (rem <= 0) is not `valid` and reads elements 0..1
            const float2 qa = *reinterpret_cast<const float2*>(px), qb = *reinterpret_cast<const float2*>(px + (rem > 2 ? 2 : 0));
            xnext[0] = qa.x; xnext[1] = qa.y; xnext[2] = rem > 2 ? qb.x : 0.f; xnext[3] = rem > 2 ? qb.y : 0.f;
        } else {
            const int rem = g.W - w0;          // independent of t: hoisted (a step that is not loaded reads elements 0..3 of x)
            xnext[0] = px[0]; xnext[1] = px[rem > 1 ? 1 : 0]; xnext[2] = px[rem > 2 ? 2 : 0]; xnext[3] = px[rem > 3 ? 3 : 0];
#pragma unroll
            for (int i = 1; i < 4; ++i) xnext[i] = i < rem ? xnext[i] : 0.f;
        }
    };
    load_x(t0);

    // window planes (wa, wb, wc) = dY planes (t-1, t, t+1); time tap kt uses plane t+1-kt
    auto step = [&](int t, float (&wa)[NV], float (&wb)[NV], float (&wc)[NV]) {
        if (t == 5) DTR(2);
        const bool more = t + 2 < g.T && t + 2 <= t1;                // dY plane t + 2 feeds the input plane t + 1 < t1
        fetch4<NCH, VW, MX>(gb, ch, (t + 2) * plane_o, more, rg);
        fetch4<NCH, VW, MX>(ab, ch, (t + 2) * plane_o, more, ra);
        float xv[4] = {xnext[0], xnext[1], xnext[2], xnext[3]};      // loaded one step ago, complete since the last LDS staging
        load_x(t + 1);                                               // next step's raw input, in flight during the stencil
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            read_plane(ring + (size_t)((t + 1 - t0) & 1) * g.slot, wc);
            float hin[4], dact[4], d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sv = fmaf(sc, xv[i], sh);
                const bool in = (w0 + i) < g.W;
                hin[i] = in ? fmaxf(sv, act_lo) : 0.f;                 // ReLU or none (see store_act)
                dact[i] = (in && sv > act_lo) ? 1.f : 0.f;
            }
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const float(&v)[NV] = kt == 0 ? wc : (kt == 1 ? wb : wa);
                if (STRIDE == 1) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wk = wt[kt * 9 + kh * 3 + kw];
                            float acc = dwacc[kt * 9 + kh * 3 + kw];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                d[i] = fmaf(wk, v[kh * 6 + i + 2 - kw], d[i]);
                                acc = fmaf(hin[i], v[kh * 6 + i + 2 - kw], acc);
                            }
                            dwacc[kt * 9 + kh * 3 + kw] = acc;
                        }
                    }
                } else {
                    // window slot 0: kh = par ? 0 : 1; window slot 1 (odd h only): kh = 2
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        if (sl == 0 || par) {
                            const float u0 = v[sl * 3], u1 = v[sl * 3 + 1], u2 = v[sl * 3 + 2];
#pragma unroll
                            for (int khc = 0; khc < 3; ++khc) {
                                const bool use = sl == 0 ? (khc == (par ? 0 : 1)) : (khc == 2);
                                if (use) {
                                    const float q0 = wt[kt * 9 + khc * 3], q1 = wt[kt * 9 + khc * 3 + 1], q2 = wt[kt * 9 + khc * 3 + 2];
                                    d[0] = fmaf(q1, u0, d[0]);
                                    d[1] = fmaf(q0, u1, fmaf(q2, u0, d[1]));
                                    d[2] = fmaf(q1, u1, d[2]);
                                    d[3] = fmaf(q0, u2, fmaf(q2, u1, d[3]));
                                    dwacc[kt * 9 + khc * 3] = fmaf(hin[1], u1, fmaf(hin[3], u2, dwacc[kt * 9 + khc * 3]));
                                    dwacc[kt * 9 + khc * 3 + 1] = fmaf(hin[0], u0, fmaf(hin[2], u1, dwacc[kt * 9 + khc * 3 + 1]));
                                    dwacc[kt * 9 + khc * 3 + 2] = fmaf(hin[1], u0, fmaf(hin[3], u1, dwacc[kt * 9 + khc * 3 + 2]));
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = stored(d[i] * dact[i], MX);
                s1 += o[i];
                s2 = fmaf(o[i], xv[i], s2);
            }
        }
        if (t == 5) DTR(3);
        // all loads of this step (dY plane t+2, x of step t+1) are consumed before the output store is issued
        store_dy<NCH, VW, STRIDE == 1>(ring + (size_t)((t - t0) & 1) * g.slot, ch, more, k0, k1, k2, rg, ra);
        if (t == 5) DTR(4);
        if (valid) {
            float* po = A.out + xbase + (size_t)t * g.H * g.W;
            if (MX) {
                const size_t oi = xbase + (size_t)t * g.H * g.W;
                if (VEC) {
                    stx4(A.out, oi, 1, o[0], o[1], o[2], o[3]);
                } else if (V2) {
                    stx2(A.out, oi, 1, o[0], o[1]);
                    if (w0 + 2 < g.W) stx2(A.out, oi + 2, 1, o[2], o[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (w0 + i < g.W) stx1(A.out, oi + i, 1, o[i]);
                }
            } else if (VEC) {
                *reinterpret_cast<float4*>(po) = make_float4(o[0], o[1], o[2], o[3]);
            } else if (V2) {
                *reinterpret_cast<float2*>(po) = make_float2(o[0], o[1]);
                if (w0 + 2 < g.W) *reinterpret_cast<float2*>(po + 2) = make_float2(o[2], o[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (w0 + i < g.W) po[i] = o[i];
            }
        }
        lds_barrier();
        if (t == 5) DTR(5);
    };

    float wv0[NV], wv1[NV], wv2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { wv0[i] = 0.f; wv1[i] = 0.f; wv2[i] = 0.f; }
    if (valid) read_plane(ring, wv1);                    // dY plane t0
    __syncthreads();       // the first step overwrites slot 0: every wave must have read plane t0 first
    if (t0 > 0) {          // a later T segment: dY plane t0 - 1 through the free slot 0 into the window (see dw_fwd_kernel)
        store_dy<NCH, VW, STRIDE == 1>(ring, ch, true, k0, k1, k2, rgm, ram);
        __syncthreads();
        if (valid) read_plane(ring, wv0);
        __syncthreads();
    }
    DTR(1);
    for (int t = t0; t < t1; t += 3) {
        step(t, wv0, wv1, wv2);
        if (t + 1 < t1) step(t + 1, wv1, wv2, wv0);
        if (t + 2 < t1) step(t + 2, wv2, wv0, wv1);
    }
    DTR(6);

    // reductions: per channel of the block, over its ipc items, in item order
    // rb[thread][29]: a thread's 29 values are consecutive words (stride 29 across lanes: no bank conflicts on the write), and
    // the summing lanes (consecutive k of one channel) read consecutive words; the [29][256] layout of round 2 put all lanes
    // of a wave on 2-3 banks
    float* rb = lds;    // reuse the ring
#pragma unroll
    for (int k = 0; k < 27; ++k) rb[tid * 29 + k] = valid ? dwacc[k] : 0.f;
    rb[tid * 29 + 27] = valid ? s1 : 0.f;
    rb[tid * 29 + 28] = valid ? s2 : 0.f;
    __syncthreads();
    for (int o = tid; o < g.cpb * 29; o += 256) {
        const int chn = o / 29, k = o - chn * 29;
        if (c0 + chn < g.C) {
            // four interleaved partial sums (fixed order): the LDS reads of a round are independent
            const float* rp = rb + (size_t)chn * g.ipc * 29 + k;
            float s0 = 0.f, s1_ = 0.f, s2_ = 0.f, s3 = 0.f;
            int i = 0;
            for (; i + 3 < g.ipc; i += 4) { s0 += rp[i * 29]; s1_ += rp[(i + 1) * 29]; s2_ += rp[(i + 2) * 29]; s3 += rp[(i + 3) * 29]; }
            for (; i < g.ipc; ++i) s0 += rp[i * 29];
            const float s = (s0 + s1_) + (s2_ + s3);
            const int slots = g.tiles * g.tsegs;
            const size_t row_id = ((size_t)n * g.C + c0 + chn) * slots + slot_id;
            if (k < 27) A.wpartial[(((size_t)n * slots + slot_id) * g.C + c0 + chn) * 27 + k] = s;   // [N][tiles * tsegs][C][27]
            else if (A.partial != nullptr) A.partial[row_id * 2 + (k - 27)] = s;
        }
    }
#ifdef X3D_TRACE
    if (tid == 0) {
        dtr[7] = wall_clock64();
        const size_t id = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (id < 16384) for (int i = 0; i < 8; ++i) g_dwtrace[id * 8 + i] = dtr[i];
    }
#endif
}

static size_t fwd_lds_bytes(const DwGeom& g) { return (2 * (size_t)g.slot + 512) * sizeof(float); }
static size_t bwd_lds_bytes(const DwGeom& g) {
    size_t ring = 2 * (size_t)g.slot, red = 29 * 256;
    return (ring > red ? ring : red) * sizeof(float);
}
// chunk slots per thread: staged rows inside the tensor (see make_chunks) x float4 chunks per row x channels
static int nch_for(const DwGeom& g, bool backward) {
    const int SH = backward ? g.Ho : g.H;
    int rows = 0;
    for (int tile = 0; tile < g.tiles; ++tile) {       // the same row arithmetic as make_chunks, maximum over the tiles
        int r_lo, nrows;
        dw_staged_rows(g.IH, dw_tile_row0(tile, g.TH, g.stride, backward), SH, &r_lo, &nrows);
        rows = nrows > rows ? nrows : rows;
    }
    return cdiv(g.cpb * rows * (g.WP / 4 - 2), 256);
}

}  // namespace

// The tile height depends on N and C too (whole rounds of workgroups, make_geom): the queries take the same N, C as the
// launch they size buffers for.
// statistics / weight-gradient slots per (n, c): row tiles x T segments of the launch with the same N, C, T
extern "C" int x3d_dw_tiles(int N, int C, int T, int H_out, int W_out) {
    DwGeom g = make_geom(N, C, T, H_out, W_out, 1, false);
    return g.tiles * g.tsegs;
}

extern "C" int x3d_dw_bwd_tiles(int N, int C, int T, int H, int W, int strideHW) {
    DwGeom g1 = make_geom(N, C, T, H, W, strideHW == 2 ? 2 : 1, true);
    return g1.tiles * g1.tsegs;
}

template <typename K, typename ARGS>
static int dw_launch(K kernel, const ARGS& args, const DwGeom& g, size_t ldsb, hipStream_t s) {
    dim3 grid(g.tiles * g.tsegs, cdiv(g.C, g.cpb), g.N), block(256);
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

#define DW_CASE3(KERNEL, N_, S_, U_, MX_)                                                              \
    (vw == 4 ? dw_launch(KERNEL<N_, S_, U_, 4, MX_>, ARGS_, GEO_, LDSB_, s)                             \
             : (vw == 2 ? dw_launch(KERNEL<N_, S_, U_, 2, MX_>, ARGS_, GEO_, LDSB_, s)                  \
                        : dw_launch(KERNEL<N_, S_, U_, 1, MX_>, ARGS_, GEO_, LDSB_, s)))
#define DW_CASE2(KERNEL, N_, S_, MX_) (uni ? DW_CASE3(KERNEL, N_, S_, true, MX_) : DW_CASE3(KERNEL, N_, S_, false, MX_))
#define DW_CASE(KERNEL, N_, S_) (mx ? DW_CASE2(KERNEL, N_, S_, true) : DW_CASE2(KERNEL, N_, S_, false))

#define DW_DISPATCH(KERNEL, ARGS, GEO, LDSB, BWD_)                                                    \
    do {                                                                                               \
        const auto& ARGS_ = ARGS; const DwGeom& GEO_ = GEO; const size_t LDSB_ = LDSB;                 \
        const int nch = nch_for(GEO_, BWD_);                                                           \
        const bool uni = GEO_.cpb == 1;                                                                \
        const bool no_v2 = x3d_opt(X3D_OPT_DW_NO_V2) != 0;                                             \
        const int vw = ((GEO_.W % 4 == 0) && (GEO_.Wo % 4 == 0)) ? 4                                   \
                     : (((GEO_.W % 2 == 0) && (GEO_.Wo % 2 == 0) && !no_v2) ? 2 : 1);                   \
        int rc_ = X3D_OK;                                                                              \
        if (nch > 8) { x3d_set_error("dw333: row too wide (W=%d)", GEO_.W); return X3D_EINVAL; }       \
        if (GEO_.stride == 1) {                                                                        \
            if (nch <= 1) rc_ = DW_CASE(KERNEL, 1, 1);                                                 \
            else if (nch <= 2) rc_ = DW_CASE(KERNEL, 2, 1);                                            \
            else if (nch <= 4) rc_ = DW_CASE(KERNEL, 4, 1);                                            \
            else rc_ = DW_CASE(KERNEL, 8, 1);                                                          \
        } else {                                                                                       \
            if (nch <= 1) rc_ = DW_CASE(KERNEL, 1, 2);                                                 \
            else if (nch <= 2) rc_ = DW_CASE(KERNEL, 2, 2);                                            \
            else if (nch <= 4) rc_ = DW_CASE(KERNEL, 4, 2);                                            \
            else rc_ = DW_CASE(KERNEL, 8, 2);                                                          \
        }                                                                                              \
        if (rc_ != X3D_OK) return rc_;                                                                 \
    } while (0)

// mx of the channelwise entries: all of the call's activation tensors are bf16, or none
#define DW_MX_FWD (X3D_MX_X | X3D_MX_Y)
#define DW_MX_BWD (X3D_MX_GA | X3D_MX_X | X3D_MX_Y)

extern "C" int x3d_dw333_fwd(const void* x, const float* w, void* y, int N, int C, int T, int H, int W,
                             int strideHW, const float* pre, int pre_act, float* partial, int mx, void* stream) {
    X3D_CHECK_ARG(x && w && y);
    X3D_CHECK_ARG(mx == 0 || mx == DW_MX_FWD);
    X3D_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    X3D_CHECK_ARG(pre_act == X3D_ACT_NONE || pre_act == X3D_ACT_RELU);     // x3d.py:147-150: ReLU precedes conv2
    DwFwdArgs A;
    A.x = (const float*)x; A.w = w; A.y = (float*)y; A.pre = pre; A.pre_act = pre ? pre_act : X3D_ACT_NONE; A.partial = partial;
    A.sp = nullptr; A.stiles = 0; A.S = 1; A.count = 0; A.gamma = A.beta = nullptr; A.rmean = A.rvar = nullptr;
    A.momentum = A.eps = 0.f; A.save = A.coef_out = nullptr;
    A.g = make_geom(N, C, T, H, W, strideHW, false);
    const size_t ldsb = fwd_lds_bytes(A.g);
    if (ldsb > 160 * 1024) { x3d_set_error("dw333_fwd: LDS tile too large (W=%d)", W); return X3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    x3d_note_kernel("dw_fwd_kernel");
    DW_DISPATCH(dw_fwd_kernel, A, A.g, ldsb, false);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_dw333_fwd_stats(const void* x, const float* w, void* y, int N, int C, int T, int H, int W,
                                   int strideHW, const float* spartial, int stiles, int S, int count,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float* save, float* coef_out, int pre_act,
                                   float* partial, int mx, void* stream) {
    X3D_CHECK_ARG(x && w && y && spartial && gamma && beta && save && coef_out);
    X3D_CHECK_ARG(mx == 0 || mx == DW_MX_FWD);
    X3D_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0 && H > 0 && W > 0 && stiles > 0 && count > 0);
    X3D_CHECK_ARG(S > 0 && N % S == 0 && (strideHW == 1 || strideHW == 2));
    X3D_CHECK_ARG(pre_act == X3D_ACT_NONE || pre_act == X3D_ACT_RELU);     // x3d.py:147-150: ReLU precedes conv2
    X3D_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
    DwFwdArgs A;
    A.x = (const float*)x; A.w = w; A.y = (float*)y; A.pre = nullptr; A.pre_act = pre_act; A.partial = partial;
    A.g = make_geom(N, C, T, H, W, strideHW, false);
    A.sp = spartial; A.stiles = stiles; A.S = S; A.count = count; A.gamma = gamma; A.beta = beta;
    A.rmean = running_mean; A.rvar = running_var; A.momentum = momentum; A.eps = eps; A.save = save; A.coef_out = coef_out;
    X3D_CHECK_ARG(A.g.cpb <= 16);
    const size_t ldsb = fwd_lds_bytes(A.g);
    if (ldsb > 160 * 1024) { x3d_set_error("dw333_fwd: LDS tile too large (W=%d)", W); return X3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    x3d_note_kernel("dw_fwd_kernel");
    DW_DISPATCH(dw_fwd_kernel, A, A.g, ldsb, false);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_dw333_bwd(const void* g, const void* a, const float* cb, const float* w, const void* x,
                             const float* pre, int pre_act, void* out, float* wpartial, float* partial, int N,
                             int C, int T, int H, int W, int strideHW, int mx, void* stream) {
    X3D_CHECK_ARG(g && a && cb && w && x && out && wpartial);
    X3D_CHECK_ARG(mx == 0 || mx == DW_MX_BWD);
    X3D_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0 && H > 0 && W > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    X3D_CHECK_ARG(pre_act == X3D_ACT_NONE || pre_act == X3D_ACT_RELU);     // x3d.py:147-150: ReLU precedes conv2
    DwBwdArgs A;
    A.g = (const float*)g; A.a = (const float*)a; A.cb = cb; A.w = w; A.x = (const float*)x; A.pre = pre;
    A.pre_act = pre ? pre_act : X3D_ACT_NONE;
    A.out = (float*)out; A.wpartial = wpartial; A.partial = partial;
    A.sp = nullptr; A.stiles = 0; A.count = 0; A.gamma = A.save = nullptr; A.dgamma = A.dbeta = nullptr;
    A.geo = make_geom(N, C, T, H, W, strideHW, true);
    const size_t ldsb = bwd_lds_bytes(A.geo);
    if (ldsb > 160 * 1024) { x3d_set_error("dw333_bwd: LDS tile too large (W=%d)", W); return X3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    x3d_note_kernel("dw_bwd_kernel");
    DW_DISPATCH(dw_bwd_kernel, A, A.geo, ldsb, true);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}

extern "C" int x3d_dw333_bwd_stats(const void* g, const void* a, const float* spartial, int stiles, int count,
                                   const float* gamma, const float* save, float* dgamma, float* dbeta, const float* w,
                                   const void* x, const float* pre, int pre_act, void* out, float* wpartial,
                                   float* partial, int N, int C, int T, int H, int W, int strideHW, int mx, void* stream) {
    X3D_CHECK_ARG(g && a && spartial && gamma && save && dgamma && dbeta && w && x && out && wpartial);
    X3D_CHECK_ARG(mx == 0 || mx == DW_MX_BWD);
    X3D_CHECK_ARG(N > 0 && N <= 65535 && C > 0 && T > 0 && H > 0 && W > 0 && stiles > 0 && count > 0);
    X3D_CHECK_ARG(strideHW == 1 || strideHW == 2);
    X3D_CHECK_ARG(pre_act == X3D_ACT_NONE || pre_act == X3D_ACT_RELU);     // x3d.py:147-150: ReLU precedes conv2
    DwBwdArgs A;
    A.g = (const float*)g; A.a = (const float*)a; A.cb = nullptr; A.w = w; A.x = (const float*)x; A.pre = pre;
    A.pre_act = pre ? pre_act : X3D_ACT_NONE;
    A.out = (float*)out; A.wpartial = wpartial; A.partial = partial;
    A.sp = spartial; A.stiles = stiles; A.count = count; A.gamma = gamma; A.save = save; A.dgamma = dgamma; A.dbeta = dbeta;
    A.geo = make_geom(N, C, T, H, W, strideHW, true);
    X3D_CHECK_ARG(A.geo.cpb <= 16);
    const size_t ldsb = bwd_lds_bytes(A.geo);
    if (ldsb > 160 * 1024) { x3d_set_error("dw333_bwd: LDS tile too large (W=%d)", W); return X3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    x3d_note_kernel("dw_bwd_kernel");
    DW_DISPATCH(dw_bwd_kernel, A, A.geo, ldsb, true);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
