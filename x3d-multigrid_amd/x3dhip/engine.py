"""Forward/backward schedule of the X3D trunk over the HIP kernels.

This is the host side of the hot path: it decides which kernel runs on which buffer in what
order; all arithmetic happens in libx3dhip.so.  The data flow is re-cut relative to the
reference's module graph (x3d.py:143-171, 316-331) so that every activation tensor is
touched the minimum number of times:

  forward, per bottleneck
    conv1  (pw GEMM)  reads x            writes a1 (raw)   + BN1 partial sums in its epilogue
    conv2  (dw 3x3x3) reads a1 (BN1+ReLU applied on load)  writes a2 (raw) + BN2 partial sums
    [SE runs on the pooled BN2 statistics: no pass over a2]
    conv3  (pw GEMM)  reads a2 (BN2*SE + Swish on load)    writes a3 (raw) + BN3 partial sums
    [downsample: pw GEMM stride 2 on x -> ad (raw) + its BN partial sums]
    epilogue          out = relu(BN3(a3) + (x | BNd(ad)))
  backward mirrors it; BN backward is an affine combination  d(raw) = A*g + B*raw + C  with
  per-(sample, channel) coefficients, applied on load by the consumer conv kernels.

Only raw conv outputs (a1, a2, a3, ad) and block outputs are kept for backward.
"""
import os

import torch

from . import _lib, ops
from ._lib import ACT_NONE, ACT_RELU, ACT_SWISH

BN_EPS = 1e-5        # nn.BatchNorm3d defaults used by SubBatchNorm3d (x3d.py:23-25)
BN_MOMENTUM = 0.1


class _BNRef:
    """Pointers to one SubBatchNorm3d's tensors (x3d.py:9-25)."""

    def __init__(self, mod):
        self.mod = mod

    @property
    def gamma(self):
        return self.mod.weight.data

    @property
    def beta(self):
        return self.mod.bias.data


def _w2d(p):
    return p.data.view(p.shape[0], -1)


class WeightPacks:
    """All pointwise weights of a model pre-packed into MFMA fragment order (forward and
    backward-data orientation), refreshed with ONE kernel launch per forward pass."""

    def __init__(self, model):
        import numpy as np
        from . import _lib
        L = _lib.lib()
        convs = []
        for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
            for blk in layer:
                convs += [blk.conv1.weight, blk.conv3.weight]
                if blk.downsample is not None:
                    convs.append(blk.downsample[0].weight)
        convs.append(model.conv5.weight)
        dev = convs[0].device
        assert L.x3d_pw_pack_job_bytes() == 48
        dt = np.dtype([("w", "<u8"), ("wp", "<u8"), ("M", "<i4"), ("K", "<i4"), ("ldm", "<i4"), ("ldk", "<i4"),
                       ("mtiles", "<i4"), ("kgroups", "<i4"), ("wg0", "<i4"), ("with_bf16", "<i4")])
        sizes, metas, items = [], [], []
        for w in convs:
            co, ci = w.shape[0], w.shape[1]
            for transposed in (False, True):
                M, K = (ci, co) if transposed else (co, ci)
                n = int(L.x3d_pw_pack_floats(K, M, 1 if transposed else 0))
                sizes.append(n)
                items.append(int(L.x3d_pw_pack_items(K, M, 1 if transposed else 0)))
                metas.append((w, transposed, M, K, 1 if transposed else ci, ci if transposed else 1))
        self.buf = ops._f((sum(sizes),), convs[0].data)      # (through ops._f: poison / guard modes cover the packs too)
        jobs = np.zeros(len(metas), dtype=dt)
        wg_job, off, wg = [], 0, 0
        self.views, self.ptrs = {}, []
        for j, ((w, transposed, M, K, ldm, ldk), n) in enumerate(zip(metas, sizes)):
            view = self.buf[off:off + n]
            self.views[(id(w), transposed)] = view
            nwg = (items[j] + 255) // 256
            jobs[j] = (w.data_ptr(), view.data_ptr(), M, K, ldm, ldk, (M + 15) // 16, (K + 15) // 16, wg,
                       3)      # bf16 planes behind the fp32 image: hi / mid / lo in both orientations
            wg_job += [j] * nwg
            self.ptrs.append(w.data_ptr())
            off += n
            wg += nwg
        self.nwg = wg
        self.jobs = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev)
        self.wg_job = torch.tensor(wg_job, dtype=torch.int32, device=dev)
        self.params = convs

    def stale(self):
        return any(w.data_ptr() != p for w, p in zip(self.params, self.ptrs[::2]))

    def refresh(self):
        from . import _lib
        _lib.check(_lib.lib().x3d_pw_pack_batch(self.jobs.data_ptr(), self.wg_job.data_ptr(), self.nwg, _lib.stream()))

    def get(self, w, transposed=False):
        return self.views[(id(w), transposed)]


def weight_packs(model):
    wp = getattr(model, "_x3d_weight_packs", None)
    if wp is None or wp.stale():
        wp = WeightPacks(model)
        model._x3d_weight_packs = wp
    return wp


class TrunkContext:
    """Everything the backward pass needs from one forward pass."""

    def __init__(self):
        self.blocks = []
        self.stem = None
        self.head = None


class _Config:
    """Host-side A/B switches of the schedule.  Attributes, read at call time: a test flips one
    in-process (`engine.cfg.no_fused_bwd = True`); each starts from the environment variable of the same meaning.  The
    switches that select between KERNELS live in the library (x3dhip._lib.set_option / include/x3dhip.h x3d_set_option)."""

    def __init__(self):
        env = os.environ.get
        self.no_dw_stats = env("X3D_NO_DW_STATS", "0") == "1"        # separate BN1 finalize launch
        self.dw_bwd_stats = env("X3D_DW_BWD_STATS", "1") == "1"      # BN2-backward finalize in the depthwise prologue
        self.no_res_fuse = env("X3D_NO_RES_FUSE", "0") == "1"        # separate bn_add_relu_bwd launch
        self.no_batch_reduce = env("X3D_NO_BATCH_REDUCE", "0") == "1"
        self.no_fused_bwd = env("X3D_NO_FUSED_BWD", "0") == "1"      # stages 1-2: separate dgrad + batched wgrad
        # X3D_WGRAD_OVERLAP=1: the postponed weight gradients of layer4 / layer3 are launched on a second stream when their
        # stage's data-gradient chain is done, beside the NEXT stage's chain.  OFF by default: measured 8.89 vs 8.59 ms per
        # step in round 2 (two forks per replay; co-running kernels slow the latency-bound chain by more than the overlap
        # hides -- the same result as round 1's per-conv side stream)
        self.wgrad_overlap = env("X3D_WGRAD_OVERLAP", "0") == "1"
        self.side_stream = env("X3D_SIDE_STREAM") == "1" and env("X3D_NO_SIDE_STREAM") != "1"
        self.exp_skip_wgrad = env("X3D_EXP_SKIP_WGRAD") == "1"       # timing experiment only: gradients are wrong
        self.no_se_merge = env("X3D_NO_SE_MERGE", "0") == "1"        # separate bn2 finalize + SE launches (round 3)


cfg = _Config()


def _fused_bwd(grads, g, x, mode=0, has_addend=False):
    """Stages 1-2: data gradient and weight gradient of a pointwise conv from one pass (ops.pw_bwd_fused)."""
    if cfg.no_fused_bwd or grads.side is not None or cfg.exp_skip_wgrad:
        return False
    if _lib.get_option("dgrad_f32") or _lib.get_option("wgrad_f32"):      # exact-fp32 A/B switches: separate kernels
        return False
    if tuple(x.shape[2:]) != tuple(g.shape[2:]):
        return False
    return ops.pw_bwd_fused_ok(x.shape[1], g.shape[1], g[0, 0].numel(), mode, has_addend, ops.pw_bwd_fused_mx(g, x))


def _bn_train(partial, bn, S, count, want_nsum=False):
    return ops.bn_fwd_finalize(partial, S, count, bn.weight.data, bn.bias.data,
                               bn.split_bn.running_mean, bn.split_bn.running_var,
                               BN_MOMENTUM, BN_EPS, want_nsum=want_nsum)


def _bn_eval(bn, N):
    return ops.bn_eval_coef(bn.bn.running_mean, bn.bn.running_var, bn.weight.data, bn.bias.data, N, BN_EPS)


def trunk_forward(model, x, training, ctx=None):
    """x: float32 [N,3,T,H,W] on the GPU.  Returns pooled features [N, C5] (task 'class') or [N, C5, T] (task 'loc',
    x3d.py:241) -- the input of the head's fc1 (x3d.py:331-333).  ctx (TrunkContext) collects what backward needs."""
    N, _, T, H, W = x.shape
    S = model.bn1.num_splits
    packs = weight_packs(model)
    packs.refresh()                      # weights changed since the last step: one launch re-packs all
    if training and N % S != 0:
        raise ValueError("batch size %d is not divisible by num_splits %d (x3d.py:50)" % (N, S))

    # ---- stem: conv1_s (raw) -> conv1_t (raw + stats) -> bn1+relu applied lazily by consumers
    a_s = ops.stem133_fwd(x, model.conv1_s.weight.data)
    a_t, part = ops.dw5t_fwd(a_s, model.conv1_t.weight.data, want_stats=training)
    P = a_t[0, 0].numel()
    if training:
        c0, save0, _ = _bn_train(part, model.bn1, S, P)
    else:
        c0, save0 = _bn_eval(model.bn1, N), None
    if ctx is not None:
        ctx.stem = dict(x=x, a_s=a_s, a_t=a_t, c0=c0, save0=save0)

    cur_raw, cur_coef = a_t, c0          # lazy: consumers apply relu(c0 * a_t)
    for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
        for blk in layer:
            cur_raw, cur_coef = _block_forward(blk, cur_raw, cur_coef, S, training, ctx, packs, wide_dtype(model))

    # ---- conv5 / bn5 / relu / global average pool
    w5 = _w2d(model.conv5.weight)
    a5, p5 = ops.pw_fwd(cur_raw, w5, want_stats=training, wp=packs.get(model.conv5.weight))
    P5 = a5[0, 0].numel()
    if training:
        c5, save5, _ = _bn_train(p5, model.bn5, S, P5)
    else:
        c5, save5 = _bn_eval(model.bn5, N), None
    pooled = ops.bn_relu_pool_fwd(a5, c5, per_frame=getattr(model, "task", "class") == "loc")
    if ctx is not None:
        ctx.head = dict(x4=cur_raw, a5=a5, c5=c5, save5=save5, S=S, w5t=packs.get(model.conv5.weight, True))
    return pooled


def wide_dtype(model):
    """Storage type of the wide (planes = 2.25 x width, x3d.py:112-116) tensors inside every bottleneck: conv1's and
    conv2's raw outputs and their gradients.  torch.float32 (default) or torch.bfloat16 = the mixed-storage mode of
    BASELINE config 5 (`model.act_dtype`, set by generate_model(..., act_dtype=...) or assigned later; every other
    tensor, all arithmetic and all statistics stay fp32)."""
    dt = getattr(model, "act_dtype", torch.float32)
    if dt not in (torch.float32, torch.bfloat16):
        raise ValueError("act_dtype must be torch.float32 or torch.bfloat16")
    return dt


def _block_forward(blk, x_raw, x_coef, S, training, ctx, packs, wide=torch.float32):
    N = x_raw.shape[0]
    pre_act = ACT_RELU if x_coef is not None else ACT_NONE
    stride = blk.stride
    w1, w3 = _w2d(blk.conv1.weight), _w2d(blk.conv3.weight)
    # conv1's output is the first wide tensor; conv2 keeps its input's storage type, conv3 returns to fp32
    a1, p1 = ops.pw_fwd(x_raw, w1, pre=x_coef, pre_act=pre_act, want_stats=training, wp=packs.get(blk.conv1.weight),
                        out_dtype=wide)
    P1 = a1[0, 0].numel()
    if training and not cfg.no_dw_stats:
        # bn1's finalize runs inside the depthwise kernel's prologue (one launch less per block)
        a2, p2, c1, s1 = ops.dw333_fwd_stats(a1, blk.conv2.weight.data, p1, S, P1, blk.bn1.weight.data, blk.bn1.bias.data,
                                             blk.bn1.split_bn.running_mean, blk.bn1.split_bn.running_var,
                                             stride=stride, pre_act=ACT_RELU, momentum=BN_MOMENTUM, eps=BN_EPS)
    else:
        if training:
            c1, s1, _ = _bn_train(p1, blk.bn1, S, P1)
        else:
            c1, s1 = _bn_eval(blk.bn1, N), None
        a2, p2 = ops.dw333_fwd(a1, blk.conv2.weight.data, stride=stride, pre=c1, pre_act=ACT_RELU,
                               want_stats=training or blk.has_se)
    P2 = a2[0, 0].numel()
    nsum2 = None
    se = None
    if training and blk.has_se and not cfg.no_se_merge and blk.bn2.weight.shape[0] <= 1024:
        # bn2's finalize and the SE branch in one launch (round 4)
        bn2 = blk.bn2
        c2e, s2, nsum2, se_v, z, pool = ops.se_bn_fwd(p2, S, P2, bn2.weight.data, bn2.bias.data, bn2.split_bn.running_mean,
                                                      bn2.split_bn.running_var, _w2d(blk.fc1.weight), blk.fc1.bias.data,
                                                      _w2d(blk.fc2.weight), blk.fc2.bias.data, BN_MOMENTUM, BN_EPS)
        se = dict(se=se_v, z=z, pool=pool, nsum=nsum2)
    else:
        if training:
            c2, s2, nsum2 = _bn_train(p2, blk.bn2, S, P2, want_nsum=blk.has_se)
        else:
            c2, s2 = _bn_eval(blk.bn2, N), None
            if blk.has_se:
                nsum2 = p2.sum(dim=2)[..., 0].contiguous()   # per-(n,c) sum of raw a2 (tiny tensor)
        if blk.has_se:
            c2e, se_v, z, pool = ops.se_fwd(c2, nsum2, P2, _w2d(blk.fc1.weight), blk.fc1.bias.data,
                                            _w2d(blk.fc2.weight), blk.fc2.bias.data)
            se = dict(se=se_v, z=z, pool=pool, nsum=nsum2)
        else:
            c2e = c2
    a3, p3 = ops.pw_fwd(a2, w3, pre=c2e, pre_act=ACT_SWISH, want_stats=training, wp=packs.get(blk.conv3.weight))
    c3 = s3 = None
    if not training:
        c3 = _bn_eval(blk.bn3, N)           # training: BN3's finalize is folded into the block-output kernel below
    ad = cd = sd = None
    if blk.downsample is not None:
        wd = _w2d(blk.downsample[0].weight)
        ad, pd = ops.pw_fwd(x_raw, wd, stride=stride, pre=x_coef, pre_act=pre_act, want_stats=training,
                            wp=packs.get(blk.downsample[0].weight))
        if training:
            cd, sd, _ = _bn_train(pd, blk.downsample[1], S, P2)
        else:
            cd, sd = _bn_eval(blk.downsample[1], N), None
        res, rcoef = ad, cd
    else:
        if x_coef is not None:
            raise RuntimeError("identity residual needs a materialised block input")
        res, rcoef = x_raw, None
    if training:
        bn3 = blk.bn3
        out, s3 = ops.bn_stats_add_relu_fwd(a3, p3, S, P2, bn3.weight.data, bn3.bias.data, bn3.split_bn.running_mean,
                                            bn3.split_bn.running_var, res, rcoef, momentum=BN_MOMENTUM, eps=BN_EPS)
    else:
        out = ops.bn_add_relu_fwd(a3, c3, res, rcoef)
    if ctx is not None:
        # transposed packs for the backward-data GEMMs (weights are unchanged until the optimizer step)
        ctx.blocks.append(dict(blk=blk, x_raw=x_raw, x_coef=x_coef, a1=a1, c1=c1, s1=s1, a2=a2, c2e=c2e, s2=s2,
                               se=se, a3=a3, s3=s3, ad=ad, sd=sd, out=out, S=S,
                               w1t=packs.get(blk.conv1.weight, True), w3t=packs.get(blk.conv3.weight, True),
                               wdt=packs.get(blk.downsample[0].weight, True)
                               if blk.downsample is not None else None))
    return out, None


class _GradSink:
    """Where parameter gradients go.  Default: fresh tensors handed back to autograd (which
    accumulates them into .grad).  Direct mode (set by x3dhip.trainer.Trainer, which owns a
    zeroed flat gradient buffer): kernels write straight into the existing .grad storage and
    autograd gets None -- 300+ tiny accumulate kernels per step disappear."""

    def __init__(self, direct, side=None):
        self.direct = direct
        self.written = {}
        self.side = side          # HIP stream for the weight-gradient kernels (off the critical path)
        # group sums of the weight-gradient partials, postponed to one launch per backward part (single-stream mode)
        self.deferred = ops.DeferredGrads() if (side is None and not cfg.no_batch_reduce) else None

    def flush(self):
        if self.deferred is not None:
            self.deferred.flush()
        self.join()

    def flush_async(self, device):
        """Launch what has been postponed so far on the second stream (fork here, join in flush()).  Every tensor the
        launches read or write stays referenced until the join."""
        d = self.deferred
        if d is None or not cfg.wgrad_overlap or self.side is not None or (not d.wjobs and not d.reduces):
            return
        main = torch.cuda.current_stream()
        st = side_stream(device)
        st.wait_stream(main)
        self._async_keep = getattr(self, "_async_keep", [])
        self._async_keep.append((d.keep, list(d.reduces)))
        with torch.cuda.stream(st):
            d.flush()
        self._async_side = st

    def join(self):
        st = getattr(self, "_async_side", None)
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)
            self._async_side = None
            self._async_keep = []

    def out(self, p):
        if self.direct and p.grad is not None:
            return p.grad.view(-1)
        return None

    def put(self, p, t):
        self.written[p] = t


def trunk_backward(model, ctx, dpooled, grads, part="all", state=None):
    """dpooled: gradient w.r.t. trunk_forward's result ([N, C5], or [N, C5, T] for task 'loc').  Fills
    grads.written[param] for every trunk parameter (grads: _GradSink).

    part = "all": the whole backward.  "late": head, layer4, layer3 only -- returns the state for
    part = "early" (layer2, layer1, stem), so that the gradient exchange of the late parameters (x3dhip/trainer.py,
    first bucket) can run while the early layers' backward executes.  Every part ends with the side stream
    joined: when it returns, all gradients of its parameters are ordered before later work on the current stream."""
    n_late = len(model.layer3) + len(model.layer4)
    blocks = list(reversed(ctx.blocks))
    if part in ("all", "late"):
        hd = ctx.head
        S = hd["S"]
        a5 = hd["a5"]
        P5 = a5[0, 0].numel()
        g5, pp = ops.bn_relu_pool_bwd(a5, hd["c5"], dpooled.contiguous())
        _bn_bwd(grads, pp, S, P5, model.bn5, hd["save5"], out_cb=True)
        cb5 = grads.last_cb
        _wgrad(grads, model.conv5.weight, g5, a5, cb5, hd["x4"])
        if _res_fusable(blocks[0]):
            dcur = ops.pw_bwd_data_res(g5, a5, cb5, _w2d(model.conv5.weight), blocks[0]["out"], blocks[0]["a3"],
                                       wpt=hd["w5t"])
        else:
            dcur, _ = ops.pw_bwd_data(g5, a5, cb5, _w2d(model.conv5.weight), wpt=hd["w5t"])
        del g5
        pstem = None
        last = len(blocks) if part == "all" else n_late
        n4 = len(model.layer4)
        for i in range(last):
            dcur, pstem = _block_backward(blocks[i], dcur, grads, blocks[i + 1] if i + 1 < len(blocks) else None)
            if i + 1 == n4 or (i + 1 == n_late and part == "all"):
                grads.flush_async(dcur[0].device if isinstance(dcur, tuple) else dcur.device)
        if part == "late":
            grads.flush()
            if grads.side is not None:
                torch.cuda.current_stream().wait_stream(grads.side)
            return dcur, pstem
    else:
        dcur, pstem = state
        for i in range(n_late, len(blocks)):
            dcur, pstem = _block_backward(blocks[i], dcur, grads, blocks[i + 1] if i + 1 < len(blocks) else None)

    S = ctx.head["S"]
    st = ctx.stem
    a_t = st["a_t"]
    P = a_t[0, 0].numel()
    _bn_bwd(grads, pstem, S, P, model.bn1, st["save0"])
    cb0 = grads.last_cb
    w_t, w_s = model.conv1_t.weight, model.conv1_s.weight
    dx_s, dwt = ops.dw5t_bwd(dcur, a_t, cb0, w_t.data, st["a_s"], dw_out=grads.out(w_t))
    grads.put(w_t, dwt)
    grads.put(w_s, ops.stem133_bwd_weight(st["x"], dx_s, w_s.shape, out=grads.out(w_s)))
    grads.flush()
    if grads.side is not None:
        torch.cuda.current_stream().wait_stream(grads.side)
    return None


def _bn_bwd(grads, partial, S, count, bn, save, out_cb=True):
    cb, dg, db = ops.bn_bwd_finalize(partial, S, count, bn.weight.data, save, dgamma=grads.out(bn.weight),
                                     dbeta=grads.out(bn.bias))
    grads.put(bn.weight, dg)
    grads.put(bn.bias, db)
    grads.last_cb = cb
    return cb


_side_streams = {}


def use_side_stream():
    """Weight-gradient kernels on a second HIP stream: OFF by default.  It paid while the data-gradient chain was slow;
    since the persistent dgrad / split-bf16 wgrad kernels the forked graph replays 3.8 % slower than the linear one
    (10.85 vs 10.46 ms at config 2: the overlapped kernels slow each other down by about what the overlap hides, and
    every fork / join is a cross-queue dependency).  X3D_SIDE_STREAM=1 turns it back on."""
    return cfg.side_stream


def side_stream(device):
    """One extra HIP stream per device for work that is off the backward critical path."""
    st = _side_streams.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[device] = st
    return st


def _wgrad(grads, w, g, a, cb, x, **kw):
    """Pointwise weight gradient.  Nothing downstream in the backward pass consumes it, so it runs
    on the side stream, concurrently with the data-gradient chain (the small stage-3/4 kernels
    cannot fill 256 CUs on their own); joined once at the end of trunk_backward."""
    if cfg.exp_skip_wgrad:      # timing experiment only: gradients are wrong
        o = grads.out(w)
        grads.put(w, o if o is not None else torch.zeros_like(w))
        return
    side = grads.side
    if side is None:
        grads.put(w, ops.pw_bwd_weight(g, a, cb, x, w.shape, out=grads.out(w), defer=grads.deferred, **kw))
        return
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw = ops.pw_bwd_weight(g, a, cb, x, w.shape, out=grads.out(w), **kw)
    for t in (g, a, cb, x, kw.get("pre")):
        if t is not None:
            t.record_stream(side)
    grads.put(w, dw)


def _on_side(grads, fn, tensors):
    """Run fn() on the side stream after everything enqueued so far on the current stream (same
    protocol as _wgrad); `tensors` are its inputs allocated on the main stream."""
    side = grads.side
    if side is None:
        return fn()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        r = fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    return r


def _res_fusable(rec):
    """The residual-add + ReLU backward of this block can ride in the epilogue of the data gradient that produces its
    output gradient (no downsample branch: one statistics pair)."""
    return rec is not None and rec["ad"] is None and not cfg.no_res_fuse


def _block_backward(rec, dout, grads, below=None):
    """dout: gradient of the block's output -- or, when the producer already applied this block's residual-add + ReLU
    backward (_res_fusable), the pair (g3, p3).  `below` is the record of the block whose output is this block's input
    (None for the first block)."""
    blk, S = rec["blk"], rec["S"]
    a1, a2, a3, ad = rec["a1"], rec["a2"], rec["a3"], rec["ad"]
    x_raw, x_coef = rec["x_raw"], rec["x_coef"]
    pre_act = ACT_RELU if x_coef is not None else ACT_NONE
    P1, P2 = a1[0, 0].numel(), a2[0, 0].numel()

    if isinstance(dout, tuple):
        (g3, p3), pd = dout, None
    else:
        g3, p3, pd = ops.bn_add_relu_bwd(dout, rec["out"], a3, ad)
    cb3 = _bn_bwd(grads, p3, S, P2, blk.bn3, rec["s3"])

    # conv3: weight gradient, then data gradient fused with the swish backward (one pass at stages 1-2)
    if _fused_bwd(grads, g3, a2, mode=1):
        w3 = blk.conv3.weight
        ds, ps, dw3 = ops.pw_bwd_fused(g3, a3, cb3, w3.shape, rec["w3t"], a2, xpre=rec["c2e"], xact=ACT_SWISH, mode=1,
                                       dw_out=grads.out(w3), defer=grads.deferred)
        grads.put(w3, dw3)
    else:
        _wgrad(grads, blk.conv3.weight, g3, a3, cb3, a2, pre=rec["c2e"], pre_act=ACT_SWISH)
        ds, ps = ops.pw_bwd_data(g3, a3, cb3, _w2d(blk.conv3.weight), x=a2, pre=rec["c2e"], pre_act=ACT_SWISH,
                                 wpt=rec["w3t"], out_dtype=a2.dtype)
    if blk.has_se:
        se = rec["se"]
        outs = None
        if grads.direct and blk.bn2.weight.grad is not None:
            outs = dict(dgamma=blk.bn2.weight.grad, dbeta=blk.bn2.bias.grad, dw1=blk.fc1.weight.grad.view(-1),
                        db1=blk.fc1.bias.grad, dw2=blk.fc2.weight.grad.view(-1), db2=blk.fc2.bias.grad)
        cb2, o = ops.se_bn_bwd_finalize(ps, S, P2, blk.bn2.weight.data, blk.bn2.bias.data, rec["s2"], se["nsum"],
                                        _w2d(blk.fc1.weight), _w2d(blk.fc2.weight), se["se"], se["z"], se["pool"],
                                        outs=outs)
        grads.put(blk.bn2.weight, o["dgamma"])
        grads.put(blk.bn2.bias, o["dbeta"])
        grads.put(blk.fc1.weight, o["dw1"])
        grads.put(blk.fc1.bias, o["db1"])
        grads.put(blk.fc2.weight, o["dw2"])
        grads.put(blk.fc2.bias, o["db2"])
    elif S == 1 and cfg.dw_bwd_stats:
        cb2 = None          # bn2's backward finalize runs in the depthwise kernel's prologue
    else:
        cb2 = _bn_bwd(grads, ps, S, P2, blk.bn2, rec["s2"])

    # conv2 (channelwise): fused data + weight backward, relu backward of bn1 in its epilogue
    bn2 = None
    if cb2 is None:
        dg2, db2 = grads.out(blk.bn2.weight), grads.out(blk.bn2.bias)
        if dg2 is None:
            dg2, db2 = ops._f(blk.bn2.weight.shape, blk.bn2.weight.data), ops._f(blk.bn2.bias.shape, blk.bn2.bias.data)
        bn2 = (ps, P2, blk.bn2.weight.data, rec["s2"], dg2, db2)
        grads.put(blk.bn2.weight, dg2)
        grads.put(blk.bn2.bias, db2)
    g1, wpart2, p1 = ops.dw333_bwd(ds, a2, cb2, blk.conv2.weight.data, a1, stride=blk.stride, pre=rec["c1"],
                                   pre_act=ACT_RELU, reduce=False, bn=bn2)
    del ds
    # the group sum of the 27-tap partials feeds only the optimizer: side stream
    w2 = blk.conv2.weight
    grads.put(w2, _on_side(grads, lambda: ops.dw333_bwd_reduce(wpart2, w2.shape, grads.out(w2), defer=grads.deferred),
                           (wpart2,)))
    cb1 = _bn_bwd(grads, p1, S, P1, blk.bn1, rec["s1"])

    # conv1 (+ downsample branch)
    mode1 = 2 if _res_fusable(below) else (1 if x_coef is not None else 0)
    fuse1 = _fused_bwd(grads, g1, x_raw, mode=mode1, has_addend=True)
    if not fuse1:
        _wgrad(grads, blk.conv1.weight, g1, a1, cb1, x_raw, pre=x_coef, pre_act=pre_act)
    if blk.downsample is not None:
        dsc, dsb = blk.downsample[0], blk.downsample[1]
        cbd = _bn_bwd(grads, pd, S, P2, dsb, rec["sd"])
        _wgrad(grads, dsc.weight, g3, ad, cbd, x_raw, stride=blk.stride, pre=x_coef, pre_act=pre_act)
        addend, _ = ops.pw_bwd_data(g3, ad, cbd, _w2d(dsc.weight), wpt=rec["wdt"])
        astride = blk.stride
    else:
        addend, astride = g3, 1
    if fuse1:
        w1 = blk.conv1.weight
        if _res_fusable(below):            # x_raw IS below["out"]: the mask of the producer's ReLU and the conv's input
            mode, kw = 2, dict(ex=below["a3"])
        elif x_coef is not None:           # lazily normalised input (layer1.0: the stem's BN + ReLU)
            mode, kw = 1, dict(xpre=x_coef, xact=pre_act)
        else:
            mode, kw = 0, {}
        dprev, pprev, dw1 = ops.pw_bwd_fused(g1, a1, cb1, w1.shape, rec["w1t"], x_raw, mode=mode, addend=addend,
                                             addend_stride=astride, dw_out=grads.out(w1), defer=grads.deferred, **kw)
        grads.put(w1, dw1)
        return ((dprev, pprev), None) if mode == 2 else (dprev, pprev)
    if _res_fusable(below):
        return ops.pw_bwd_data_res(g1, a1, cb1, _w2d(blk.conv1.weight), below["out"], below["a3"], addend=addend,
                                   addend_stride=astride, wpt=rec["w1t"]), None
    dprev, pprev = ops.pw_bwd_data(g1, a1, cb1, _w2d(blk.conv1.weight), x=x_raw if x_coef is not None else None,
                                   pre=x_coef, pre_act=pre_act, addend=addend, addend_stride=astride, wpt=rec["w1t"])
    return dprev, pprev


def trunk_parameters(model):
    """Trunk parameters in a fixed order (everything except the head's fc1/fc2)."""
    out = []
    for name, p in model.named_parameters():
        if name.startswith("fc1.") or name.startswith("fc2."):
            continue
        out.append(p)
    return out


class TrunkFunction(torch.autograd.Function):
    """autograd node for the whole trunk (same convention as the reference's only custom op,
    SwishEfficient, x3d.py:71-84: forward(ctx, ...) / backward(ctx, grad))."""

    @staticmethod
    def forward(ctx, model, x, *params):
        tctx = TrunkContext()
        pooled = trunk_forward(model, x, True, tctx)
        ctx.model = model
        ctx.tctx = tctx
        ctx.params = params
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        import os
        side = side_stream(dpooled.device) if use_side_stream() else None
        sink = _GradSink(getattr(ctx.model, "_direct_grads", False), side)
        trunk_backward(ctx.model, ctx.tctx, dpooled, sink)
        ctx.tctx = None
        out = []
        for p in trunk_parameters(ctx.model):
            t = sink.written[p]
            if sink.direct and p.grad is not None and t.data_ptr() == p.grad.data_ptr():
                out.append(None)              # already written in place
            else:
                out.append(t.view(p.shape))
        return (None, None) + tuple(out)
