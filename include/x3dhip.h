/*
 * x3dhip.h -- C ABI of libx3dhip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * X3D training hot path.
 *
 * The reference (KiyoshiKAWASAKI/X3D-Multigrid) has no FFI of its own: every FLOP of
 * x3d.py runs inside PyTorch (ATen/cuDNN/oneDNN).  The boundary replaced here is therefore
 * the set of ATen ops x3d.py calls, re-cut so that elementwise work rides in the prologue
 * or epilogue of the convolution that produces/consumes it.  Each entry point cites the
 * reference call site(s) it replaces (file:line in /root/reference).
 *
 * Conventions
 *   - all tensors fp32, NCTHW contiguous (x3d.py:316), P = T*H*W
 *   - plain pointers and sizes; the caller (torch) owns every buffer; no hidden allocation,
 *     no hidden synchronisation; every kernel is enqueued on the hipStream_t passed as
 *     `stream` (void* in this header so that C callers need no HIP headers)
 *   - return 0 on success, negative X3D_E* on failure; x3d_last_error() gives the message
 *     (thread-local); no C++ exception crosses the boundary; thread-safe, re-entrant
 *   - "coef" arrays are per-(sample, channel): the split-BN's interleaved sample->split map
 *     (x3d.py:50-52) is resolved when they are built, so conv kernels never see splits
 *   - "partial" arrays hold per-workgroup partial reductions that a finalize kernel sums in
 *     fixed order in fp64 (bitwise reproducible; no float atomics anywhere)
 */
#ifndef X3DHIP_H
#define X3DHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X3D_ABI_VERSION 7

#define X3D_OK 0
#define X3D_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define X3D_ELAUNCH (-2)  /* hipLaunch error */

/* activation selectors for fused prologues / epilogues */
#define X3D_ACT_NONE 0
#define X3D_ACT_RELU 1   /* nn.ReLU           x3d.py:119,148,169,210 */
#define X3D_ACT_SWISH 2  /* SwishEfficient    x3d.py:71-84 */

/* Mixed-storage mode (ABI 5; BASELINE config 5 "bf16 storage / fp32 accumulate"): the `mx` argument of the pointwise and
 * channelwise entry points says which ACTIVATION tensors of that call are stored as bf16 (2 bytes per element, same
 * [N][C][T][H][W] order) instead of fp32 -- in the model these are the wide (planes = 2.25 x width) tensors inside a
 * bottleneck (x3d.py:112-116: conv1 output, conv2 output and their gradients).  Everything else (weights, packs, BN
 * coefficients, statistics partials, weight-gradient partials, narrow activations) is fp32; every product is accumulated
 * in fp32; a bf16 store rounds to nearest even.  mx = 0 is the fp32 path, bitwise what ABI 4 computed.  A combination an
 * entry point has no kernel for returns X3D_EINVAL (never a silent fallback). */
#define X3D_MX_X 1    /* the forward input x of the convolution (also where it is re-read by the backward entries) */
#define X3D_MX_Y 2    /* the tensor the call writes: forward y, backward out / dx */
#define X3D_MX_GA 4   /* the upstream gradient g AND the raw forward output a it is combined with */

int x3d_abi_version(void);
const char* x3d_last_error(void);
/* measurement aid: name of the kernel template (e.g. "pw6_kernel", "dw_bwd_kernel") the last pointwise / channelwise entry
 * point called from this thread launched -- the names a rocprofv3 --kernel-trace summary prints (bench.py groups by it) */
const char* x3d_last_kernel(void);
/* debug aid (tests): one launch that fills the LDS of every CU with a NaN pattern, so that a kernel which consumes an LDS
 * word it never wrote produces a NaN instead of a plausible value; `sink`: any device buffer of >= 4 bytes (never written) */
int x3d_debug_poison_lds(void* sink, void* stream);

/* Tuning / A-B options (ABI 6).  The reference has no counterpart (its only knobs are the module constants of
 * train_x3d_kinetics_multigrid.py:49-61); these select between kernels of this library that compute the same function.
 * Options are process-wide integers, read at CALL time by every entry point and tile-count query (so a query and the
 * launch it sizes buffers for must not straddle a change); each starts from its default or from the environment variable
 * of the same meaning (X3D_FB_GRID, X3D_DGRAD_F32, ...: DESIGN.md section 7).  Names: x3d_option_name(0 .. x3d_option_count()-1):
 *   fb_grid, pw_pgrid, pw_nt4_min, pw_no_persist, dw_th, dw_balance, dw_no_v2, no_pw6, no_pw7, no_pwfs, dgrad_f32,
 *   wgrad_f32, bwd_terms (3 = fp32-level three-term bf16 split of the backward GEMM operands, 2 = two-term, ~2^-16),
 *   no_wgrad4, wg_cpw, wg_cap, stem_wg_cap, dw_tsplit_wgs (channelwise launches of at most this many workgroups cut the T
 *   march into two segments; 0 = never; backward kernel), dw_tsplit_wgs_fwd (the same for the forward kernel), dw_cpb_max (channels per channelwise workgroup, <= 16),
 *   pw6_min_m (smallest output-channel count the whole-K forward kernel takes), pw_two_tiles_k (padded K from which a wave
 *   of the whole-K kernels takes two M tiles of one staged tile).
 * Unknown name or out-of-range value: X3D_EINVAL. */
int x3d_set_option(const char* name, int value);
int x3d_get_option(const char* name, int* value);
int x3d_reset_options(void);
int x3d_option_count(void);
const char* x3d_option_name(int index);

/* ------------------------------------------------------------------------------------
 * Pointwise 1x1x1 convolution  (conv1x1x1 x3d.py:98-103; Bottleneck.conv1/conv3 :112,116;
 * downsample[0] :272; conv5 :231) as an fp32 MFMA GEMM  Y[co,p] = sum_ci W[co,ci] * in[ci,p].
 * ---------------------------------------------------------------------------------- */

/* number of voxel tiles per sample the pw kernels use for `partial`: N samples, K = reduction
 * channels (forward: Cin, backward-data: Cout), M = channels of the tensor the partial sums
 * describe (forward: Cout, backward-data: Cin), P = its T*H*W, dense = 0 for the strided
 * (downsample) forward, else 1 */
int x3d_pw_tiles(int N, int K, int M, int P, int dense);
/* tiles of x3d_pw_fwd's `partial` (packed = 1 when wpacked is passed: the large-channel layers then run on 32-voxel items) */
int x3d_pw_fwd_tiles(int N, int Cin, int Cout, int P, int dense, int packed);
/* tiles of x3d_pw_bwd_data's / x3d_pw_bwd_data_res's `partial` (packed = 1 when wpacked_t is passed; mx as in the call) */
int x3d_pw_bwd_tiles(int N, int Cin, int Cout, int P, int packed, int mx);

/* Forward.  in[ci,p] = act(pre[n,ci,0] * x + pre[n,ci,1]) when pre != NULL (fuses the
 * producer's BN-apply + ReLU, or BN-apply * SE-scale + Swish: x3d.py:147-148,151-160), else x.
 * strideHW in {1,2}: 2 = the downsample conv's (1,2,2) stride (x3d.py:101), x is [N,Cin,T,H,W]
 * and y is [N,Cout,T,Ho,Wo].  partial (may be NULL) receives per-(n,co,tile) {sum y, sum y^2}
 * as float[N][Cout][x3d_pw_tiles(N,Cin,Cout,Po,strideHW==1)][2] for the BN that follows (x3d.py:51). */
/* Weight pre-packing for the tiled variant used on large-C layers (K >= 64 and M >= 96, see
 * x3d_pw_wants_packed): x3d_pw_pack writes w[Cout][Cin] into MFMA fragment order
 * (x3d_pw_pack_floats(K, M, transposed) floats, zero padded); transposed = 1 packs the backward-data
 * operand (M = Cin, K = Cout).  Behind the fp32 image every pack carries THREE bf16 planes in 16x16x32 MFMA
 * fragment order (hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid): all 24 significant bits) -- in
 * both orientations since ABI 6; the split-precision forward, data-gradient, fused and weight-gradient
 * kernels read them (a two-term kernel reads hi and mid).
 * Weights change every optimizer step: pack once per step.
 * Passing NULL for wpacked selects the streaming kernel (same results, same `partial` shape). */
int x3d_pw_wants_packed(int K, int M);
size_t x3d_pw_pack_floats(int K, int M, int transposed);
size_t x3d_pw_pack_items(int K, int M, int transposed);   /* work items (256 per workgroup) of one batched pack job */
int x3d_pw_pack(const float* w, float* wp, int Cout, int Cin, int transposed, void* stream);
/* Batched packing: `jobs` is a device array of records
 *   { const float* w; float* wp; int M, K, ldm, ldk, mtiles, kgroups, wg0, with_bf16; }   (x3d_pw_pack_job_bytes() each;
 *   A[row][k] = w[row*ldm + k*ldk], mtiles = ceil(M/16), kgroups = ceil(K/16), wg0 = first workgroup of the job;
 *   with_bf16: 0 = fp32 image only, any other value = the three bf16 planes as well -- since ABI 6 the kernels
 *   need them: pass 3.  x3d_pw_pack_floats / _items always size a pack for three planes)
 * and wg_job[n_workgroups] maps every 256-element workgroup to its job. */
size_t x3d_pw_pack_job_bytes(void);
int x3d_pw_pack_batch(const void* jobs, const int* wg_job, int n_workgroups, void* stream);

int x3d_pw_fwd(const void* x, const float* w, const float* wpacked, void* y,
               int N, int Cin, int Cout, int T, int H, int W, int strideHW,
               const float* pre, int pre_act,
               float* partial, int mx /* X3D_MX_X | X3D_MX_Y; stride 1 with wpacked only */, void* stream);

/* Backward-data (autograd of the conv above, fused with the BN backward that precedes it in
 * the backward pass and with the activation backward that follows it):
 *   dY[co,p]  = cb[n,co,0]*g[co,p] + cb[n,co,1]*a[co,p] + cb[n,co,2]    (BN backward, see
 *               x3d_bn_bwd_finalize; g = upstream gradient, a = raw conv output saved in fwd)
 *   dIn[ci,p] = sum_co W[co,ci] * dY[co,p]  (+ addend[ci,p'] when addend != NULL; with
 *               addend_stride 2 the addend is [N,Cin,T,Ho,Wo] and only lands on even (h,w))
 *   out       = dIn                                     when pre == NULL
 *             = dIn * act'(pre[n,ci,0]*x + pre[n,ci,1]) when pre != NULL (x = raw producer
 *               output; SwishEfficient.backward x3d.py:80-84 / ReLU backward)
 * partial (NULL unless pre != NULL) gets per-(n,ci,tile) {sum out, sum out*x}.
 * Geometry: g,a are [N,Cout,T,H,W]; out,x are [N,Cin,T,H,W]. */
int x3d_pw_bwd_data(const void* g, const void* a, const float* cb, const float* w,
                    const float* wpacked_t,
                    void* out, int N, int Cin, int Cout, int T, int H, int W,
                    const void* x, const float* pre, int pre_act,
                    const float* addend, int addend_stride,
                    float* partial, int mx /* X3D_MX_GA | X3D_MX_X (x) | X3D_MX_Y (out); with wpacked_t only */,
                    void* stream);

/* The same data gradient with the residual-add + ReLU backward of the block that PRODUCED this conv's input folded into
 * the epilogue (x3d.py:165-169 backward; what x3d_bn_add_relu_bwd does in a launch of its own for blocks without a
 * downsample branch):
 *   out[ci,p] = (dIn[ci,p] + addend) where res_out[ci,p] > 0, else 0      (= g3 of that block)
 *   partial   = per-(n,ci,tile) {sum out, sum out*res_raw}                 (its bn3 backward statistics)
 * res_out = that block's output [N,Cin,T,H,W], res_raw = its raw conv3 output. */
int x3d_pw_bwd_data_res(const void* g, const void* a, const float* cb, const float* w, const float* wpacked_t,
                        float* out, int N, int Cin, int Cout, int T, int H, int W, const float* res_out,
                        const float* res_raw, const float* addend, int addend_stride, float* partial,
                        int mx /* X3D_MX_GA only */, void* stream);

/* Fused backward (stages 1-2): the data gradient AND the weight-gradient partials of one pointwise convolution from ONE
 * pass over g, a and x -- ConvolutionBackward's grad_input + grad_weight of conv1x1x1 (x3d.py:98-103 as Bottleneck.conv1 /
 * conv3, :146,162) with the same fused BN backward in front and the same epilogues behind as the two separate entry
 * points above (split-bf16 MFMA, 3 products, fp32 accumulate):
 *   dY = cb0*g + cb1*a + cb2;   dIn[ci,p] = sum_co W[co,ci] dY[co,p] (+ addend);   dW[co,ci] += sum_p dY[co,p] X[ci,p]
 *   X  = act(xpre0*x + xpre1) (xpre != NULL) or x -- the convolution's forward input
 *   mode 0: out = dIn
 *   mode 1: out = dIn * act'(xpre0*x + xpre1)                 partial {sum out, sum out*x}      (x3d_pw_bwd_data with pre)
 *   mode 2: out = dIn where x > 0 else 0 (x = the producing block's output, xpre == NULL)
 *                                                             partial {sum out, sum out*ex}     (x3d_pw_bwd_data_res)
 * wpacked_t = the transposed pack of x3d_pw_pack (its split-bf16 planes are the data gradient's A operand).
 * wpartial is float[x3d_pw_bwd_fused_groups(N,P)][Cout][Cin] (x3d_reduce_partials sums it); partial is
 * float[N][Cin][x3d_pw_bwd_fused_tiles(N,P)][2].  x3d_pw_bwd_fused_ok tells whether (Cin, Cout, P, mode, addend) is in the kernel's set
 * (dense, P % 4 == 0, both channel counts <= 128); other shapes use the separate entry points. */
/* mx: 0; X3D_MX_GA (g, a bf16: any mode); X3D_MX_X | X3D_MX_Y (x and dx bf16: mode 1 without addend) */
int x3d_pw_bwd_fused_ok(int Cin, int Cout, int P, int mode, int has_addend, int mx);
int x3d_pw_bwd_fused_groups(int N, int P);
int x3d_pw_bwd_fused_tiles(int N, int P);   /* (round 4: one slot per workgroup touching a sample, not per 64-voxel chunk) */
int x3d_pw_bwd_fused(const void* g, const void* a, const float* cb, const float* wpacked_t, const void* x,
                     const float* xpre, int xact, int mode, const float* ex, const float* addend, int addend_stride,
                     void* dx, float* wpartial, float* partial, int N, int Cin, int Cout, int T, int H, int W,
                     int mx, void* stream);

/* Backward-weight: dW[co,ci] = sum_{n,p} dY[co,p] * in[ci,p] with dY and in formed as above
 * (strideHW 2: in is sampled at even (h,w) of x[N,Cin,T,H,W]; g,a are at output resolution).
 * wpartial is float[x3d_pw_wgrad_groups(...)][Cout][Cin]; x3d_reduce_partials sums it. */
int x3d_pw_wgrad_groups(int N, int P, int Cout, int Cin, int strideHW);
int x3d_pw_bwd_weight(const void* g, const void* a, const float* cb,
                      const void* x, const float* pre, int pre_act,
                      float* wpartial, int N, int Cin, int Cout, int T, int H, int W,
                      int strideHW, int mx /* X3D_MX_GA | X3D_MX_X; the MFMA tile variants only */, void* stream);

/* Several weight gradients in as few launches as there are tile variants among them (jobs: the arguments of
 * x3d_pw_bwd_weight, one struct per conv).  Nothing in the backward pass consumes a weight gradient, so the host may
 * postpone them all to the end of the pass; results are bitwise those of the single calls. */
typedef struct X3DWgradJob {
    const void* g; const void* a; const float* cb; const void* x; const float* pre; float* wpartial;
    int pre_act, N, Cin, Cout, T, H, W, strideHW, mx;
} X3DWgradJob;
size_t x3d_wgrad_job_bytes(void);
int x3d_pw_bwd_weight_batch(const X3DWgradJob* jobs, int njobs, void* stream);

/* out[i] = sum_g partial[g][i], g < groups, i < n  (fixed order, fp64 accumulate) */
int x3d_reduce_partials(const float* partial, float* out, int groups, int n, void* stream);

/* The same reduction for `njobs` independent (partial, out, groups, n) jobs in one launch (host arrays of length njobs;
 * the weight-gradient group sums of a whole backward pass).  Bitwise identical to njobs x3d_reduce_partials calls. */
int x3d_reduce_partials_batch(const float* const* partials, float* const* outs, const int* groups, const int* ns,
                              int njobs, void* stream);

/* ------------------------------------------------------------------------------------
 * Channelwise 3x3x3 convolution (conv3x3x3 x3d.py:87-95, Bottleneck.conv2 :114,150):
 * groups=C, pad 1, stride (1,s,s), no bias.  HBM-bound; LDS-staged T-marching stencil.
 * ---------------------------------------------------------------------------------- */
int x3d_dw_tiles(int N, int C, int T, int H_out, int W_out);  /* slots per (n,c) of `partial`: row tiles x T segments (same N, C, T as the launch; ABI 6) */

/* y = dw333(relu(pre*x+pre) zero-padded).  partial: float[N][C][tiles][2] {sum y, sum y^2}.
 * pre_act of the channelwise entries: X3D_ACT_RELU or X3D_ACT_NONE only (x3d.py:147-150: ReLU precedes conv2);
 * X3D_ACT_SWISH returns X3D_EINVAL. */
int x3d_dw333_fwd(const void* x, const float* w, void* y,
                  int N, int C, int T, int H, int W, int strideHW,
                  const float* pre, int pre_act, float* partial, int mx /* 0 or X3D_MX_X | X3D_MX_Y */, void* stream);

/* Training form with the producer BN's finalize folded in (one launch less per block): the scale / shift applied while
 * loading x are derived inside the kernel from the producer conv's statistics partials
 * spartial float[N][C][stiles][2] (x3d.py:47-58; S splits, `count` voxels per (n, c), eps, running statistics with
 * momentum / unbiased variance).  coef_out float[N][C][2] and save float[2][S][C] (mean, invstd) are written for the
 * backward pass. */
int x3d_dw333_fwd_stats(const void* x, const float* w, void* y, int N, int C, int T, int H, int W, int strideHW,
                        const float* spartial, int stiles, int S, int count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* save,
                        float* coef_out, int pre_act, float* partial, int mx /* 0 or X3D_MX_X | X3D_MX_Y */,
                        void* stream);

/* Fused backward (data + weight) of the conv above:
 *   dY = cb0*g + cb1*a + cb2 (g,a at output resolution [N,C,T,Ho,Wo])
 *   hin = act(pre*x+pre)  (x raw [N,C,T,H,W])
 *   out = dw333^T(dY) * act'(pre*x+pre)           -> [N,C,T,H,W]
 *   dW[c,kt,kh,kw] partials: float[N][x3d_dw_bwd_tiles(N,C,T,H,W,s)][C][27] (group-sum over the first two dims)
 *   partial: float[N][C][x3d_dw_bwd_tiles(N,C,T,H,W,s)][2] {sum out, sum out*x}  */
int x3d_dw_bwd_tiles(int N, int C, int T, int H, int W, int strideHW);   /* (ABI 6: takes T, as x3d_dw_tiles) */
int x3d_dw333_bwd(const void* g, const void* a, const float* cb, const float* w,
                  const void* x, const float* pre, int pre_act,
                  void* out, float* wpartial, float* partial,
                  int N, int C, int T, int H, int W, int strideHW,
                  int mx /* 0 or X3D_MX_GA | X3D_MX_X | X3D_MX_Y: g, a, x and out are the same four wide tensors */,
                  void* stream);

/* The same with the producer BN's backward finalize folded in (single split, num_splits == 1): the coefficients are
 * derived inside the kernel from spartial float[N][C][stiles][2] = {sum g, sum g*a} (the statistics the kernel that
 * wrote g left), gamma and save = {mean[C], invstd[C]}; dgamma / dbeta float[C] are written (not accumulated). */
int x3d_dw333_bwd_stats(const void* g, const void* a, const float* spartial, int stiles, int count,
                        const float* gamma, const float* save, float* dgamma, float* dbeta, const float* w,
                        const void* x, const float* pre, int pre_act, void* out, float* wpartial, float* partial,
                        int N, int C, int T, int H, int W, int strideHW, int mx, void* stream);

/* ------------------------------------------------------------------------------------
 * Stem (x3d.py:196-208,317-318): dense 1x3x3 s(1,2,2) 3->C, then depthwise temporal 5x1x1.
 * ---------------------------------------------------------------------------------- */
int x3d_stem133_fwd(const float* x, const float* w, float* y,
                    int N, int Cin, int Cout, int T, int H, int W, void* stream);
/* dW[co,ci,kh,kw] partials float[x3d_stem_wgrad_groups(N,T)][Cout][Cin*9] from dy[N,Cout,T,Ho,Wo] */
int x3d_stem_wgrad_groups(int N, int T);
int x3d_stem133_bwd_weight(const float* x, const float* dy, float* wpartial,
                           int N, int Cin, int Cout, int T, int H, int W, void* stream);
/* y = dw5t(x) (pad (2,0,0)); partial float[N][C][x3d_dw5t_tiles(HW)][2] {sum y, sum y^2} */
int x3d_dw5t_tiles(int HW);
int x3d_dw5t_fwd(const float* x, const float* w, float* y, int N, int C, int T, int HW,
                 float* partial, void* stream);
/* dY = cb0*g+cb1*a+cb2; dx = dw5t^T(dY); dW partials float[N][tiles][C][5] (x = conv1_s output) */
int x3d_dw5t_bwd(const float* g, const float* a, const float* cb, const float* w, const float* x,
                 float* dx, float* wpartial, int N, int C, int T, int HW, void* stream);

/* ------------------------------------------------------------------------------------
 * Split BatchNorm3d (SubBatchNorm3d x3d.py:9-58) finalisation kernels (tiny).
 * ---------------------------------------------------------------------------------- */

/* bytes of device scratch the finalize entry points below need (fp64 tile sums etc.) */
size_t x3d_finalize_scratch_bytes(int N, int C, int Wd);

/* Training forward: reduce partial[N][C][tiles][2] over the samples of each split
 * (sample n belongs to split n % S, x3d.py:50), produce
 *   coef[n][c] = {gamma*invstd, beta - mean*gamma*invstd}  (BN + affine, x3d.py:51-57)
 *   save[0][j][c] = mean, save[1][j][c] = invstd           (float[2][S][C])
 *   nsum[n][c]   = sum_p raw[n,c,p]                        (for SE pooling / its backward)
 *   running_mean/var[j*C+c] updated with momentum 0.1, unbiased variance (nn.BatchNorm3d).
 * count = elements per (n,c) = T*H*W of the normalised tensor. */
int x3d_bn_fwd_finalize(const float* partial, int N, int C, int tiles, int S, int count,
                        const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps,
                        float* coef, float* save, float* nsum, void* scratch, void* stream);

/* Eval forward: coef from the aggregated running stats (x3d.py:54, bn branch). */
int x3d_bn_eval_coef(const float* running_mean, const float* running_var,
                     const float* gamma, const float* beta, float eps,
                     int N, int C, float* coef, void* stream);

/* Squeeze-excitation forward on pooled statistics (x3d.py:153-159): for every sample
 *   pool[c] = coef[n][c][0]*nsum[n][c]/count + coef[n][c][1]   (global average of BN output)
 *   z = relu(W1 pool + b1); se = sigmoid(W2 z + b2)
 *   coef_out[n][c] = coef[n][c] * se[c]      (so the consumer's prologue computes swish(bn*se))
 * save_se: float[N][C] (se), save_z: float[N][Wd] (post-relu hidden). */
int x3d_se_fwd(const float* coef, const float* nsum, int N, int C, int Wd, int count,
               const float* w1, const float* b1, const float* w2, const float* b2,
               float* coef_out, float* save_se, float* save_z, float* save_pool, void* stream);

/* Training forward of a Bottleneck with SE: x3d_bn_fwd_finalize (bn2, x3d.py:151) and x3d_se_fwd (x3d.py:153-159) in ONE
 * launch (ABI 7).  partial[N][C][tiles][2] are conv2's statistics tiles; outputs as the two calls: save[2][S][C],
 * nsum[N][C], save_se / save_pool[N][C], save_z[N][Wd], running statistics updated, and coef_out[n][c] = bn2's
 * {scale, shift} * se[n][c] (what conv3's prologue applies before its Swish).  C <= 1024, Wd <= 64. */
int x3d_se_bn_fwd(const float* partial, int N, int C, int tiles, int S, int count,
                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                  float momentum, float eps, int Wd, const float* w1, const float* b1, const float* w2,
                  const float* b2, float* coef_out, float* save, float* nsum, float* save_se, float* save_z,
                  float* save_pool, void* stream);

/* BN backward finalisation.  partial[N][C][tiles][2] = {sum g, sum g*raw} per (n,c,tile) where
 * g is the gradient w.r.t. the BN+affine output.  Produces
 *   cb[n][c] = {A, B, Cc} with d(raw) = A*g + B*raw + Cc   (nn.BatchNorm3d training backward)
 *   dgamma[c] (+)= sum g*xhat, dbeta[c] (+)= sum g      (accumulate != 0 adds to existing) */
int x3d_bn_bwd_finalize(const float* partial, int N, int C, int tiles, int S, int count,
                        const float* gamma, const float* save,
                        float* cb, float* dgamma, float* dbeta, int accumulate,
                        void* scratch, void* stream);

/* BN2 + SE backward finalisation (Bottleneck with SE).  partial = {sum ds, sum ds*raw} where
 * ds is the gradient w.r.t. s = bn2(raw)*se (already through swish').  Computes the SE
 * gradients (dW1,db1,dW2,db2), and cb such that d(raw) = A*ds + B*raw + Cc. */
int x3d_se_bn_bwd_finalize(const float* partial, int N, int C, int tiles, int S, int count,
                           const float* gamma, const float* beta, const float* save,
                           const float* nsum,
                           int Wd, const float* w1, const float* w2,
                           const float* save_se, const float* save_z, const float* save_pool,
                           float* cb, float* dgamma, float* dbeta,
                           float* dw1, float* db1, float* dw2, float* db2,
                           void* scratch, void* stream);

/* ------------------------------------------------------------------------------------
 * Residual epilogue (x3d.py:165-169): out = relu(c3*a3 + res) with res = x (identity) or
 * cd*ad (downsample branch BN).  Backward: g = dout*(out>0), partial {sum g, sum g*a3}
 * (and {sum g, sum g*ad} in partial_d when ad != NULL).
 * ---------------------------------------------------------------------------------- */
int x3d_ew_tiles(int P);
int x3d_bn_add_relu_fwd(const float* a3, const float* c3, const float* res, const float* cd,
                        float* out, int N, int C, int P, void* stream);
/* Training form with BN3's finalize folded in: scale/shift are derived inside the kernel from conv3's
 * statistics partials float[N][C][tiles][2] (x3d.py:47-58: per-(split, channel) mean / biased variance over
 * N/S samples x count voxels, eps, running statistics with momentum and unbiased variance);
 * save float[2][S][C] receives (mean, invstd) for the backward pass.  cd as above. */
int x3d_bn_stats_add_relu_fwd(const float* a3, const float* partial, int tiles, int S, int count,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* save, const float* res, const float* cd,
                              float* out, int N, int C, int P, void* stream);
int x3d_bn_add_relu_bwd(const float* dout, const float* out, const float* a3, const float* ad,
                        float* g, float* partial, float* partial_d,
                        int N, int C, int P, void* stream);

/* Head pooling (x3d.py:328-331): pooled[n][c][s] = mean over segment s of relu(c5*a5); segs = 1 is
 * AdaptiveAvgPool3d((1,1,1)) (task 'class', x3d.py:239), segs = T is AdaptiveAvgPool3d((None,1,1)) (task 'loc',
 * x3d.py:241; segment = one T plane, P % segs == 0).  Backward: g = dpooled[n][c][s]/(P/segs) * (c5*a5 > 0),
 * partial {sum g, sum g*a5}. */
int x3d_bn_relu_pool_fwd(const float* a5, const float* c5, float* pooled,
                         int N, int C, int P, int segs, void* stream);
int x3d_bn_relu_pool_bwd(const float* a5, const float* c5, const float* dpooled,
                         float* g, float* partial, int N, int C, int P, int segs, void* stream);

/* ------------------------------------------------------------------------------------
 * Classification head (x3d.py:333-343: fc1 = 1x1x1 conv on the pooled vector without bias, ReLU, Dropout, fc2 = Linear)
 * and the training script's loss (nn.CrossEntropyLoss, mean over rows: train_x3d_kinetics_multigrid.py:189,259).
 * R rows = N (task 'class') or N*T (task 'loc'); pooled[R][K], w1[J][K], w2[C][J], b2[C]; J % 4 == 0, K <= 640.
 * rng = device {seed, draw counter} (two uint64) for the dropout mask (hash of seed, counter, element); x3d_head_ce /
 * x3d_head_advance_rng advance the counter on the device, so replayed hipGraphs draw fresh masks.
 * ---------------------------------------------------------------------------------- */
size_t x3d_head_scratch_floats(int R, int K, int J, int C);
/* hd[R][J] = dropout(relu(pooled W1^T)) (kept for the backward), logits[R][C] = hd W2^T + b2 */
int x3d_head_fwd(const float* pooled, const float* w1, const float* w2, const float* b2, float* hd, float* logits,
                 int R, int K, int J, int C, float p_drop, const unsigned long long* rng, void* stream);
/* loss[0] = mean_r CE(logits[r], labels[r]); dlogits = d loss / d logits; scratch >= R floats */
int x3d_head_ce(const float* logits, const long long* labels, float* loss, float* dlogits, float* scratch, int R, int C,
                unsigned long long* rng, void* stream);
int x3d_head_advance_rng(unsigned long long* rng, float* dummy1, void* stream);
/* dW1[J][K], dW2[C][J], db2[C], dpooled[R][K] from dlogits[R][C]; scratch: x3d_head_scratch_floats floats */
int x3d_head_bwd(const float* dlogits, const float* hd, const float* pooled, const float* w1, const float* w2,
                 float* dw1, float* dw2, float* db2, float* dpooled, float* scratch, int R, int K, int J, int C,
                 float p_drop, void* stream);

/* Charades localisation losses (train_x3d_charades_loc.py:123,168-189) on the per-frame logits [B][C][T] of task 'loc'
 * (x3d.py:340-343): linear interpolation to the label length TL (F.interpolate, align_corners False), then
 * losses[0] = cls_loss = BCEWithLogits(max_t, max_t labels), losses[1] = loc_loss = BCEWithLogits(per frame);
 * dlogits = d[(cls + loc) * grad_scale] / d logits (the script: grad_scale = 1 / (2 num_steps_per_update)).
 * labels are float [B][C][TL]; scratch >= 2*B*C floats. */
int x3d_loc_losses(const float* logits, const float* labels, float* losses, float* dlogits, float* scratch, int B, int C,
                   int T, int TL, float grad_scale, void* stream);

/* Stand-alone SubBatchNorm3d.forward (x3d.py:47-58 used as a module of its own; inside the network the statistics ride in
 * conv epilogues): x3d_bn_rowstats writes partial[N][C][x3d_ew_tiles(P)][2] = {sum x, sum x^2} (g == NULL) or
 * {sum g, sum g*x} for x3d_bn_fwd_finalize / x3d_bn_bwd_finalize; x3d_bn_affine applies the per-(sample, channel) map
 * out = c0*x + c1 (ncoef 2: coef[N][C][2]) or out = c0*g + c1*x + c2 (ncoef 3: the BN backward). */
int x3d_bn_rowstats(const float* x, const float* g, float* partial, int N, int C, int P, void* stream);
int x3d_bn_affine(const float* x, const float* g, const float* coef, float* out, int N, int C, int P, int ncoef, void* stream);

/* Gradient accumulation over micro-batches (`loss = cls_loss / num_steps_per_update; loss.backward()` repeated
 * num_steps_per_update times before optimizer.step(), train_x3d_kinetics_multigrid.py:119,267-273):
 * acc = (first ? 0 : acc) + scale * g over the flat gradient buffer. */
int x3d_grad_accumulate(float* acc, const float* g, size_t n, float scale, int first, void* stream);

/* Fused SGD (torch.optim.SGD, train_x3d_kinetics_multigrid.py:183): g += wd*w;
 * m = first ? g : mu*m + g; w -= lr*m.  grad_scale multiplies g first (1/world_size). */
int x3d_sgd_fused(float* w, const float* g, float* m, size_t n, float lr, float momentum,
                  float weight_decay, float grad_scale, int first, void* stream);

/* ------------------------------------------------------------------------------------
 * GPU-side clip input pipeline.  Replaces, per sample, the CPU work of
 * kinetics_multigrid.py:240-253 on decoded uint8 frames: frame selection (TemporalRandomCrop,
 * transforms/temporal_transforms.py:94-117 -- indices computed by the host), crop + PIL bilinear
 * resize (MultiScaleRandomCropMultigrid, transforms/spatial_transforms.py:480-495), horizontal flip
 * (:334-346), ToTensor(255) + Normalize (:44-83,106-116), stack/permute to [3][T][S][S].
 * Bit-exact with Pillow's 8-bit bilinear resample given the host-built coefficient table
 * (x3dhip/clip_input.py:resize_coeffs = Resample.c precompute_coeffs + normalize_coeffs_8bpc).
 * `jobs`: device array of X3DClipJob, one per sample; all pointers are device pointers. */
typedef struct {
    const unsigned char* src;   /* [Tsrc][Hs][Ws][3] decoded frames */
    unsigned char* tmp;         /* scratch [T][crop][out][3] */
    float* dst;                 /* [3][T][out][out]: the sample's slice of the NCTHW batch */
    const int* kk;              /* [out][ksize] 22-bit fixed-point coefficients */
    const int* bounds;          /* [out][2] (first input index, tap count) */
    const int* frames;          /* [T] 0-based source frame per output frame */
    int Hs, Ws, x1, y1, crop, out, ksize, T, flip, pad;
} X3DClipJob;
size_t x3d_clip_job_bytes(void);
int x3d_clip_preprocess(const void* jobs, int njobs, int max_T, int max_crop, int max_out,
                        const float* mean3, const float* std3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* X3DHIP_H */
