"""Tensor-level wrappers over the C ABI (include/x3dhip.h).

PyTorch is used for device memory and streams only: every function below allocates its outputs
with torch, passes raw pointers to libx3dhip.so on the current HIP stream and returns
tensors.  There is no fallback: a missing library or a CPU tensor raises.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SWISH, check, ptr  # noqa: F401


def _need_cuda(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise _lib.X3DHipError("x3dhip ops need CUDA(HIP) tensors; got a CPU tensor "
                                       "(the product path has no CPU fallback)")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.X3DHipError("x3dhip ops need contiguous float32 tensors")


# Mixed-storage mode (include/x3dhip.h X3D_MX_*): the wide tensors inside a bottleneck may be torch.bfloat16; the flags
# below are derived from the tensors' dtypes, so a call is fp32 exactly when all of its tensors are.
MX_X, MX_Y, MX_GA = 1, 2, 4


def _need_act(*ts):
    """Activation tensors: contiguous CUDA(HIP) float32 or bfloat16."""
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise _lib.X3DHipError("x3dhip ops need CUDA(HIP) tensors; got a CPU tensor "
                                       "(the product path has no CPU fallback)")
            if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous():
                raise _lib.X3DHipError("x3dhip ops need contiguous float32 (or, mixed-storage mode, bfloat16) activations")


def _bf(t):
    return t is not None and t.dtype == torch.bfloat16


def _same_dtype(name, *ts):
    ts = [t for t in ts if t is not None]
    if any(t.dtype != ts[0].dtype for t in ts):
        raise _lib.X3DHipError("%s: tensors that share a storage flag must have one dtype (got %s)"
                               % (name, [str(t.dtype) for t in ts]))


def out_hw(h, stride):
    return (h - 1) // 2 + 1 if stride == 2 else h


# Poison mode (debug / tests): every buffer this module allocates is filled with NaN before the kernel that owns it
# runs, so a slot a kernel is supposed to write but skips -- statistics partials, weight-gradient partials, pack pads --
# surfaces as a NaN in a named tensor instead of as whatever the caching allocator left there.  Read at call time
# (`set_poison`, or X3D_POISON=1 in the environment at import); off by default: the product path never pays the fills.
_poison = os.environ.get("X3D_POISON", "0") == "1"


def set_poison(on):
    """Turn NaN-filling of every freshly allocated buffer on / off; returns the previous setting."""
    global _poison
    prev, _poison = _poison, bool(on)
    return prev


def poisoned():
    return _poison


# Guard mode (debug / tests): every buffer is allocated with GUARD elements of a canary pattern in front of and behind it;
# `check_guards()` lists the buffers whose canaries a kernel has overwritten.  An out-of-bounds WRITE lands in whatever
# tensor the caching allocator happened to place next to the victim -- i.e. its effect depends on what ran earlier in the
# process -- and no parity test of the kernel's own output can see it; the canaries make it deterministic.
GUARD = 1024
_CANARY = {4: 0x5EEDBEEF - (1 << 32) if 0x5EEDBEEF >= (1 << 31) else 0x5EEDBEEF, 2: 0x5EED, 1: 0x5E, 8: 0x5EEDBEEF5EEDBEEF}
_INT_OF = {4: torch.int32, 2: torch.int16, 1: torch.uint8, 8: torch.int64}
_guard = False
_guarded = []


def set_guard(on):
    """Turn guard-band allocation on / off (drops the record of earlier guarded buffers); returns the previous setting."""
    global _guard
    prev, _guard = _guard, bool(on)
    del _guarded[:]
    return prev


def check_guards():
    """[(shape, dtype, 'front' | 'back', first overwritten guard element)] of every guarded buffer a kernel wrote outside of."""
    bad = []
    for raw, n, shape, dtype in _guarded:
        iv = raw.view(_INT_OF[raw.element_size()])
        c = _CANARY[raw.element_size()]
        for name, seg in (("front", iv[:GUARD]), ("back", iv[GUARD + n:])):
            hit = (seg != c).nonzero()
            if hit.numel():
                bad.append((tuple(shape), str(dtype), name, int(hit[0])))
    return bad


def _f(shape, like, dtype=torch.float32):
    if _guard:
        n = 1
        for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
            n *= int(d)
        raw = torch.empty(n + 2 * GUARD, dtype=dtype, device=like.device)
        raw.view(_INT_OF[raw.element_size()]).fill_(_CANARY[raw.element_size()])
        t = raw[GUARD:GUARD + n].view(shape)
        if _poison:
            t.fill_(float("nan"))
        _guarded.append((raw, n, shape, dtype))
        return t
    t = torch.empty(shape, dtype=dtype, device=like.device)
    if _poison:
        t.fill_(float("nan"))
    return t


_scratch = {}
_scratch_retired = []


def scratch(dev, nbytes):
    """Grow-only per-device scratch for the finalize kernels (fp64 tile sums).  Its raw pointer is baked into captured
    hipGraphs, so a block that is outgrown is RETIRED, not freed: graphs captured earlier keep replaying into memory that
    is still theirs (a few hundred KB per retired block; `release_retired_scratch` drops them once no graph is left)."""
    cur = _scratch.get(dev)
    if cur is None or cur.numel() < nbytes:
        if cur is not None:
            _scratch_retired.append(cur)
        cur = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        if _poison:
            cur.fill_(0xFF)              # all-ones bytes read back as NaN in fp32 and fp64
        _scratch[dev] = cur
    return cur


def release_retired_scratch():
    """Called by Trainer.invalidate_graphs() after every captured graph has been dropped."""
    del _scratch_retired[:]


def finalize_scratch(dev, N, C, Wd=0):
    return scratch(dev, _lib.lib().x3d_finalize_scratch_bytes(N, C, max(Wd, 1)))


# ----------------------------------------------------------------------------- pointwise
def pw_pack(w, transposed=False):
    """Pre-pack a [Cout, Cin] weight for the tiled pw kernel (None when the layer is too small
    to use it).  transposed=True packs the backward-data operand."""
    L = _lib.lib()
    Cout, Cin = w.shape
    K, M = (Cout, Cin) if transposed else (Cin, Cout)
    if not L.x3d_pw_wants_packed(K, M):
        return None
    wp = _f((L.x3d_pw_pack_floats(K, M, 1 if transposed else 0),), w)
    check(L.x3d_pw_pack(ptr(w), ptr(wp), Cout, Cin, 1 if transposed else 0, _lib.stream()))
    return wp


def pw_fwd(x, w, stride=1, pre=None, pre_act=ACT_NONE, want_stats=True, out=None, partial=None, wp=None,
           out_dtype=torch.float32):
    """out_dtype=torch.bfloat16 (or a bfloat16 x): mixed-storage mode, see X3D_MX_* in include/x3dhip.h."""
    _need_cuda(w, pre)
    _need_act(x, out)
    L = _lib.lib()
    N, Cin, T, H, W = x.shape
    Cout = w.shape[0]
    Ho, Wo = out_hw(H, stride), out_hw(W, stride)
    y = out if out is not None else _f((N, Cout, T, Ho, Wo), x, out_dtype)
    mx = (MX_X if _bf(x) else 0) | (MX_Y if _bf(y) else 0)
    if want_stats and partial is None:
        partial = _f((N, Cout, L.x3d_pw_fwd_tiles(N, Cin, Cout, T * Ho * Wo, 1 if stride == 1 else 0,
                                                  1 if wp is not None else 0), 2), x)
    check(L.x3d_pw_fwd(ptr(x), ptr(w), ptr(wp), ptr(y), N, Cin, Cout, T, H, W, stride, ptr(pre), pre_act,
                       ptr(partial) if want_stats else None, mx, _lib.stream()))
    return y, (partial if want_stats else None)


def pw_bwd_data(g, a, cb, w, x=None, pre=None, pre_act=ACT_NONE, addend=None, addend_stride=1,
                out=None, partial=None, wpt=None, out_dtype=torch.float32):
    _need_cuda(cb, w, pre, addend)
    _need_act(g, a, x, out)
    _same_dtype("pw_bwd_data", g, a)
    L = _lib.lib()
    N, Cout, T, H, W = g.shape
    Cin = w.shape[1]
    o = out if out is not None else _f((N, Cin, T, H, W), g, out_dtype)
    mx = (MX_GA if _bf(g) else 0) | (MX_X if _bf(x) else 0) | (MX_Y if _bf(o) else 0)
    if pre is not None and partial is None:
        partial = _f((N, Cin, L.x3d_pw_bwd_tiles(N, Cin, Cout, T * H * W, 1 if wpt is not None else 0, mx), 2), g)
    check(L.x3d_pw_bwd_data(ptr(g), ptr(a), ptr(cb), ptr(w), ptr(wpt), ptr(o), N, Cin, Cout, T, H, W, ptr(x), ptr(pre),
                            pre_act, ptr(addend), addend_stride, ptr(partial) if pre is not None else None, mx,
                            _lib.stream()))
    return o, (partial if pre is not None else None)


def pw_bwd_data_res(g, a, cb, w, res_out, res_raw, addend=None, addend_stride=1, wpt=None):
    """pw_bwd_data with the residual-add + ReLU backward of the block that produced this conv's input in its epilogue:
    returns (g3, partial) of that block -- what bn_add_relu_bwd(dout, out, a3) returns for a block without downsample."""
    _need_cuda(cb, w, res_out, res_raw, addend)
    _need_act(g, a)
    _same_dtype("pw_bwd_data_res", g, a)
    L = _lib.lib()
    N, Cout, T, H, W = g.shape
    Cin = w.shape[1]
    if tuple(res_out.shape) != (N, Cin, T, H, W) or tuple(res_raw.shape) != (N, Cin, T, H, W):
        raise ValueError("pw_bwd_data_res: res_out / res_raw must be [N, Cin, T, H, W]")
    o = _f((N, Cin, T, H, W), g)
    mx = MX_GA if _bf(g) else 0
    partial = _f((N, Cin, L.x3d_pw_bwd_tiles(N, Cin, Cout, T * H * W, 1 if wpt is not None else 0, mx), 2), g)
    check(L.x3d_pw_bwd_data_res(ptr(g), ptr(a), ptr(cb), ptr(w), ptr(wpt), ptr(o), N, Cin, Cout, T, H, W, ptr(res_out),
                                ptr(res_raw), ptr(addend), addend_stride, ptr(partial), mx, _lib.stream()))
    return o, partial


def reduce_partials(partial, n_out, out=None):
    L = _lib.lib()
    groups = partial.shape[0]
    o = out if out is not None else _f((n_out,), partial)
    check(L.x3d_reduce_partials(ptr(partial), ptr(o), groups, n_out, _lib.stream()))
    return o


def reduce_partials_batch(jobs):
    """jobs: list of (partial [groups, n] contiguous, out [n]) -- every out[i] = sum_g partial[g][i] in ONE launch (per
    96 jobs).  The tensors must stay alive until the launch has been enqueued (they are: the caller holds the list)."""
    if not jobs:
        return
    L = _lib.lib()
    nj = len(jobs)
    parts = (ctypes.c_void_p * nj)(*[p.data_ptr() for p, _ in jobs])
    outs = (ctypes.c_void_p * nj)(*[o.data_ptr() for _, o in jobs])
    groups = (ctypes.c_int * nj)(*[p.shape[0] for p, _ in jobs])
    ns = (ctypes.c_int * nj)(*[p.shape[1] for p, _ in jobs])
    for p, o in jobs:
        if not (p.is_contiguous() and o.is_contiguous() and o.numel() == p.shape[1] and p.dtype == o.dtype == torch.float32):
            raise ValueError("reduce_partials_batch: partial must be contiguous float32 [groups, n], out float32 [n]")
    check(L.x3d_reduce_partials_batch(ctypes.cast(parts, ctypes.c_void_p), ctypes.cast(outs, ctypes.c_void_p),
                                      ctypes.cast(groups, ctypes.c_void_p), ctypes.cast(ns, ctypes.c_void_p), nj,
                                      _lib.stream()))


class _WgradJob(ctypes.Structure):          # X3DWgradJob (include/x3dhip.h)
    _fields_ = [("g", ctypes.c_void_p), ("a", ctypes.c_void_p), ("cb", ctypes.c_void_p), ("x", ctypes.c_void_p),
                ("pre", ctypes.c_void_p), ("wpartial", ctypes.c_void_p), ("pre_act", ctypes.c_int), ("N", ctypes.c_int),
                ("Cin", ctypes.c_int), ("Cout", ctypes.c_int), ("T", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("strideHW", ctypes.c_int), ("mx", ctypes.c_int)]


class DeferredGrads:
    """Weight-gradient work of a backward pass, postponed to a few launches at its end (nothing downstream consumes a
    weight gradient): `wjobs` -- pointwise weight-gradient kernels (x3d_pw_bwd_weight_batch), `reduces` -- the group sums
    of their partials and of the channelwise convs' (x3d_reduce_partials_batch).  Holds every tensor it points to."""

    def __init__(self):
        self.wjobs, self.keep, self.reduces = [], [], []

    def flush(self):
        L = _lib.lib()
        if self.wjobs:
            if L.x3d_wgrad_job_bytes() != ctypes.sizeof(_WgradJob):
                raise RuntimeError("X3DWgradJob layout mismatch between libx3dhip and x3dhip.ops")
            arr = (_WgradJob * len(self.wjobs))(*self.wjobs)
            check(L.x3d_pw_bwd_weight_batch(ctypes.cast(arr, ctypes.c_void_p), len(self.wjobs), _lib.stream()))
        reduce_partials_batch(self.reduces)
        self.wjobs, self.keep, self.reduces = [], [], []


def pw_bwd_weight(g, a, cb, x, w_shape, stride=1, pre=None, pre_act=ACT_NONE, out=None, wpartial=None, defer=None):
    _need_cuda(cb, pre)
    _need_act(g, a, x)
    _same_dtype("pw_bwd_weight", g, a)
    mx = (MX_GA if _bf(g) else 0) | (MX_X if _bf(x) else 0)
    L = _lib.lib()
    N, Cin, T, H, W = x.shape
    Cout = g.shape[1]
    Ho, Wo = out_hw(H, stride), out_hw(W, stride)
    groups = L.x3d_pw_wgrad_groups(N, T * Ho * Wo, Cout, Cin, stride)
    if wpartial is None:
        wpartial = _f((groups, Cout, Cin), g)
    if defer is not None:                 # kernel and group sum postponed to the batched launches (DeferredGrads.flush)
        o = out if out is not None else _f((Cout * Cin,), g)
        defer.wjobs.append(_WgradJob(ptr(g), ptr(a), ptr(cb), ptr(x), ptr(pre), ptr(wpartial), pre_act, N, Cin, Cout, T,
                                     H, W, stride, mx))
        defer.keep.append((g, a, cb, x, pre, wpartial))
        defer.reduces.append((wpartial.view(groups, -1), o))
        return o.view(w_shape)
    check(L.x3d_pw_bwd_weight(ptr(g), ptr(a), ptr(cb), ptr(x), ptr(pre), pre_act, ptr(wpartial), N, Cin, Cout, T,
                              H, W, stride, mx, _lib.stream()))
    dw = reduce_partials(wpartial.view(groups, -1), Cout * Cin, out=out)
    return dw.view(w_shape)


def pw_bwd_fused_ok(Cin, Cout, P, mode=0, has_addend=False, mx=0):
    return bool(_lib.lib().x3d_pw_bwd_fused_ok(Cin, Cout, P, mode, 1 if has_addend else 0, mx))


def pw_bwd_fused_mx(g, x):
    """Storage flags of a fused backward call with upstream gradient g and forward input x (dx takes x's dtype)."""
    return (MX_GA if _bf(g) else 0) | ((MX_X | MX_Y) if _bf(x) else 0)


def pw_bwd_fused(g, a, cb, w_shape, wpt, x, xpre=None, xact=ACT_NONE, mode=0, ex=None, addend=None, addend_stride=1,
                 dw_out=None, defer=None):
    """Data gradient + weight gradient of a pointwise conv in one pass (x3d_pw_bwd_fused).  mode 0 plain, 1 activation
    backward (x raw, xpre), 2 residual-add + ReLU backward of the producing block (x = its output, ex = its raw conv3
    output).  Returns (dx, partial or None, dW); the group sum of the dW partials goes to `defer` (DeferredGrads) when
    given, else it runs here."""
    _need_cuda(cb, wpt, xpre, ex, addend)
    _need_act(g, a, x)
    _same_dtype("pw_bwd_fused", g, a)
    L = _lib.lib()
    N, Cout, T, H, W = g.shape
    Cin = x.shape[1]
    P = T * H * W
    if tuple(x.shape) != (N, Cin, T, H, W):
        raise ValueError("pw_bwd_fused: x must be [N, Cin, T, H, W] of the gradient's geometry (dense convolution)")
    groups = L.x3d_pw_bwd_fused_groups(N, P)
    mx = pw_bwd_fused_mx(g, x)
    dx = _f((N, Cin, T, H, W), g, x.dtype)          # the gradient of a bf16 tensor is stored as bf16
    wpartial = _f((groups, Cout * Cin), g)
    partial = _f((N, Cin, L.x3d_pw_bwd_fused_tiles(N, P), 2), g) if mode != 0 else None
    check(L.x3d_pw_bwd_fused(ptr(g), ptr(a), ptr(cb), ptr(wpt), ptr(x), ptr(xpre), xact, mode, ptr(ex), ptr(addend),
                             addend_stride, ptr(dx), ptr(wpartial), ptr(partial), N, Cin, Cout, T, H, W, mx,
                             _lib.stream()))
    o = dw_out if dw_out is not None else _f((Cout * Cin,), g)
    if defer is not None:
        defer.reduces.append((wpartial, o))
        defer.keep.append((wpartial,))
    else:
        reduce_partials(wpartial, Cout * Cin, out=o)
    return dx, partial, o.view(w_shape)


# ----------------------------------------------------------------------------- channelwise
def dw333_fwd(x, w, stride=1, pre=None, pre_act=ACT_RELU, want_stats=True, out=None, partial=None):
    _need_cuda(w, pre)
    _need_act(x, out)
    _same_dtype("dw333_fwd", x, out)
    L = _lib.lib()
    N, C, T, H, W = x.shape
    Ho, Wo = out_hw(H, stride), out_hw(W, stride)
    y = out if out is not None else _f((N, C, T, Ho, Wo), x, x.dtype)
    if want_stats and partial is None:
        partial = _f((N, C, L.x3d_dw_tiles(N, C, T, Ho, Wo), 2), x)
    check(L.x3d_dw333_fwd(ptr(x), ptr(w), ptr(y), N, C, T, H, W, stride, ptr(pre), pre_act,
                          ptr(partial) if want_stats else None, (MX_X | MX_Y) if _bf(x) else 0, _lib.stream()))
    return y, (partial if want_stats else None)


def dw333_fwd_stats(x, w, spartial, S, count, gamma, beta, running_mean, running_var, stride=1, pre_act=ACT_RELU,
                    momentum=0.1, eps=1e-5):
    """dw333_fwd in training with the producer BN's finalize folded in.  Returns (y, partial, coef[N,C,2], save[2,S,C])."""
    _need_cuda(w, spartial)
    _need_act(x)
    L = _lib.lib()
    N, C, T, H, W = x.shape
    Ho, Wo = out_hw(H, stride), out_hw(W, stride)
    y = _f((N, C, T, Ho, Wo), x, x.dtype)
    partial = _f((N, C, L.x3d_dw_tiles(N, C, T, Ho, Wo), 2), x)
    coef = _f((N, C, 2), x)
    save = _f((2, S, C), x)
    check(L.x3d_dw333_fwd_stats(ptr(x), ptr(w), ptr(y), N, C, T, H, W, stride, ptr(spartial), spartial.shape[2], S, count,
                                ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), momentum, eps, ptr(save),
                                ptr(coef), pre_act, ptr(partial), (MX_X | MX_Y) if _bf(x) else 0, _lib.stream()))
    return y, partial, coef, save


def dw333_bwd(g, a, cb, w, x, stride=1, pre=None, pre_act=ACT_RELU, out=None, wpartial=None, partial=None,
              dw_out=None, reduce=True, bn=None):
    """reduce=False: returns (out, wpartial, partial) and leaves the [N*tiles] -> 1 group sum of the weight-gradient
    partials to dw333_bwd_reduce (postponed by the engine: nothing in the backward chain reads dW).
    bn = (spartial [N,C,stiles,2], count, gamma, save [2,1,C], dgamma [C], dbeta [C]) with cb = None: the producer BN's
    backward finalize (single split) runs in the kernel's prologue; dgamma / dbeta are written."""
    _need_cuda(cb, w, pre)
    _need_act(g, a, x, out)
    _same_dtype("dw333_bwd", g, a, x, out)
    mx = (MX_GA | MX_X | MX_Y) if _bf(x) else 0
    L = _lib.lib()
    N, C, T, H, W = x.shape
    tiles = L.x3d_dw_bwd_tiles(N, C, T, H, W, stride)
    o = out if out is not None else _f(x.shape, x, x.dtype)
    if wpartial is None:
        wpartial = _f((N, tiles, C, 27), x)
    if partial is None:
        partial = _f((N, C, tiles, 2), x)
    if bn is not None:
        sp, count, gamma, save, dgamma, dbeta = bn
        if cb is not None or save.numel() != 2 * C or tuple(sp.shape[:2]) != (N, C):
            raise ValueError("dw333_bwd: bn=... needs cb=None, a single BN split and spartial [N, C, tiles, 2]")
        _need_cuda(sp, gamma, save, dgamma, dbeta)
        check(L.x3d_dw333_bwd_stats(ptr(g), ptr(a), ptr(sp), sp.shape[2], count, ptr(gamma), ptr(save), ptr(dgamma),
                                    ptr(dbeta), ptr(w), ptr(x), ptr(pre), pre_act, ptr(o), ptr(wpartial), ptr(partial),
                                    N, C, T, H, W, stride, mx, _lib.stream()))
    else:
        check(L.x3d_dw333_bwd(ptr(g), ptr(a), ptr(cb), ptr(w), ptr(x), ptr(pre), pre_act, ptr(o), ptr(wpartial),
                              ptr(partial), N, C, T, H, W, stride, mx, _lib.stream()))
    if not reduce:
        return o, wpartial, partial
    return o, dw333_bwd_reduce(wpartial, w.shape, dw_out), partial


def dw333_bwd_reduce(wpartial, w_shape, dw_out=None, defer=None):
    """dW[c][27] = sum over (n, tile) groups of the [N][tiles][C][27] partials."""
    N, tiles, C, _ = wpartial.shape
    if defer is not None:
        o = dw_out if dw_out is not None else _f((C * 27,), wpartial)
        defer.reduces.append((wpartial.view(N * tiles, C * 27), o))
        return o.view(w_shape)
    dw = reduce_partials(wpartial.view(N * tiles, C * 27), C * 27, out=dw_out)
    return dw.view(w_shape)


# ----------------------------------------------------------------------------- stem
def stem133_fwd(x, w, out=None):
    _need_cuda(x, w)
    L = _lib.lib()
    N, Cin, T, H, W = x.shape
    Cout = w.shape[0]
    y = out if out is not None else _f((N, Cout, T, out_hw(H, 2), out_hw(W, 2)), x)
    check(L.x3d_stem133_fwd(ptr(x), ptr(w), ptr(y), N, Cin, Cout, T, H, W, _lib.stream()))
    return y


def stem133_bwd_weight(x, dy, w_shape, out=None):
    _need_cuda(x, dy)
    L = _lib.lib()
    N, Cin, T, H, W = x.shape
    Cout = dy.shape[1]
    groups = L.x3d_stem_wgrad_groups(N, T)
    wp = _f((groups, Cout, Cin * 9), x)
    check(L.x3d_stem133_bwd_weight(ptr(x), ptr(dy), ptr(wp), N, Cin, Cout, T, H, W, _lib.stream()))
    return reduce_partials(wp.view(groups, -1), Cout * Cin * 9, out=out).view(w_shape)


def dw5t_fwd(x, w, want_stats=True, out=None):
    _need_cuda(x, w)
    L = _lib.lib()
    N, C, T, H, W = x.shape
    y = out if out is not None else _f(x.shape, x)
    partial = _f((N, C, L.x3d_dw5t_tiles(H * W), 2), x) if want_stats else None
    check(L.x3d_dw5t_fwd(ptr(x), ptr(w), ptr(y), N, C, T, H * W, ptr(partial), _lib.stream()))
    return y, partial


def dw5t_bwd(g, a, cb, w, x, out=None, dw_out=None):
    _need_cuda(g, a, cb, w, x)
    L = _lib.lib()
    N, C, T, H, W = x.shape
    tiles = L.x3d_dw5t_tiles(H * W)
    dx = out if out is not None else _f(x.shape, x)
    wp = _f((N, tiles, C, 5), x)
    check(L.x3d_dw5t_bwd(ptr(g), ptr(a), ptr(cb), ptr(w), ptr(x), ptr(dx), ptr(wp), N, C, T, H * W, _lib.stream()))
    return dx, reduce_partials(wp.view(N * tiles, C * 5), C * 5, out=dw_out).view(w.shape)


# ----------------------------------------------------------------------------- BN / SE
def bn_fwd_finalize(partial, S, count, gamma, beta, running_mean=None, running_var=None, momentum=0.1,
                    eps=1e-5, want_nsum=False):
    L = _lib.lib()
    N, C, tiles, _ = partial.shape
    coef = _f((N, C, 2), partial)
    save = _f((2, S, C), partial)
    nsum = _f((N, C), partial) if want_nsum else None
    sc = finalize_scratch(partial.device, N, C)
    check(L.x3d_bn_fwd_finalize(ptr(partial), N, C, tiles, S, count, ptr(gamma), ptr(beta), ptr(running_mean),
                                ptr(running_var), momentum, eps, ptr(coef), ptr(save), ptr(nsum), ptr(sc),
                                _lib.stream()))
    return coef, save, nsum


def bn_eval_coef(running_mean, running_var, gamma, beta, N, eps=1e-5):
    L = _lib.lib()
    C = gamma.shape[0]
    coef = _f((N, C, 2), gamma)
    check(L.x3d_bn_eval_coef(ptr(running_mean), ptr(running_var), ptr(gamma), ptr(beta), eps, N, C, ptr(coef),
                             _lib.stream()))
    return coef


def se_fwd(coef, nsum, count, w1, b1, w2, b2):
    L = _lib.lib()
    N, C, _ = coef.shape
    Wd = w1.shape[0]
    coef_out = _f((N, C, 2), coef)
    se = _f((N, C), coef)
    z = _f((N, Wd), coef)
    pool = _f((N, C), coef)
    check(L.x3d_se_fwd(ptr(coef), ptr(nsum), N, C, Wd, count, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(coef_out),
                       ptr(se), ptr(z), ptr(pool), _lib.stream()))
    return coef_out, se, z, pool


def se_bn_fwd(partial, S, count, gamma, beta, running_mean, running_var, w1, b1, w2, b2, momentum=0.1, eps=1e-5):
    """bn_fwd_finalize (bn2) + se_fwd in one launch (training forward of a Bottleneck with SE, x3d.py:151-159).
    Returns (coef_out, save, nsum, se, z, pool)."""
    L = _lib.lib()
    N, C, tiles, _ = partial.shape
    Wd = w1.shape[0]
    coef_out = _f((N, C, 2), partial)
    save = _f((2, S, C), partial)
    nsum = _f((N, C), partial)
    se = _f((N, C), partial)
    z = _f((N, Wd), partial)
    pool = _f((N, C), partial)
    check(L.x3d_se_bn_fwd(ptr(partial), N, C, tiles, S, count, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                          momentum, eps, Wd, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(coef_out), ptr(save), ptr(nsum),
                          ptr(se), ptr(z), ptr(pool), _lib.stream()))
    return coef_out, save, nsum, se, z, pool


def bn_bwd_finalize(partial, S, count, gamma, save, dgamma=None, dbeta=None, accumulate=False):
    L = _lib.lib()
    N, C, tiles, _ = partial.shape
    cb = _f((N, C, 3), partial)
    if dgamma is None:
        dgamma, dbeta = _f((C,), partial), _f((C,), partial)
    sc = finalize_scratch(partial.device, N, C)
    check(L.x3d_bn_bwd_finalize(ptr(partial), N, C, tiles, S, count, ptr(gamma), ptr(save), ptr(cb), ptr(dgamma),
                                ptr(dbeta), 1 if accumulate else 0, ptr(sc), _lib.stream()))
    return cb, dgamma, dbeta


def se_bn_bwd_finalize(partial, S, count, gamma, beta, save, nsum, w1, w2, se, z, pool, outs=None):
    L = _lib.lib()
    N, C, tiles, _ = partial.shape
    Wd = w1.shape[0]
    cb = _f((N, C, 3), partial)
    if outs is None:
        outs = dict(dgamma=_f((C,), partial), dbeta=_f((C,), partial), dw1=_f((Wd, C), partial),
                    db1=_f((Wd,), partial), dw2=_f((C, Wd), partial), db2=_f((C,), partial))
    sc = finalize_scratch(partial.device, N, C, Wd)
    check(L.x3d_se_bn_bwd_finalize(ptr(partial), N, C, tiles, S, count, ptr(gamma), ptr(beta), ptr(save), ptr(nsum),
                                   Wd, ptr(w1), ptr(w2), ptr(se), ptr(z), ptr(pool), ptr(cb), ptr(outs["dgamma"]),
                                   ptr(outs["dbeta"]), ptr(outs["dw1"]), ptr(outs["db1"]), ptr(outs["dw2"]),
                                   ptr(outs["db2"]), ptr(sc), _lib.stream()))
    return cb, outs


# ----------------------------------------------------------------------------- elementwise
def bn_add_relu_fwd(a3, c3, res, cd=None, out=None):
    _need_cuda(a3, c3, res, cd)
    L = _lib.lib()
    N, C = a3.shape[:2]
    P = a3[0, 0].numel()
    o = out if out is not None else _f(a3.shape, a3)
    check(L.x3d_bn_add_relu_fwd(ptr(a3), ptr(c3), ptr(res), ptr(cd), ptr(o), N, C, P, _lib.stream()))
    return o


def bn_stats_add_relu_fwd(a3, partial, S, count, gamma, beta, running_mean, running_var, res, cd=None, momentum=0.1,
                          eps=1e-5, out=None):
    """Block output in training with BN3's finalize folded in.  Returns (out, save[2, S, C])."""
    _need_cuda(a3, partial, res, cd)
    L = _lib.lib()
    N, C = a3.shape[:2]
    P = a3[0, 0].numel()
    tiles = partial.shape[2]
    o = out if out is not None else _f(a3.shape, a3)
    save = _f((2, S, C), a3)
    check(L.x3d_bn_stats_add_relu_fwd(ptr(a3), ptr(partial), tiles, S, count, ptr(gamma), ptr(beta), ptr(running_mean),
                                      ptr(running_var), momentum, eps, ptr(save), ptr(res), ptr(cd), ptr(o), N, C, P,
                                      _lib.stream()))
    return o, save


def bn_add_relu_bwd(dout, out, a3, ad=None, g=None):
    _need_cuda(dout, out, a3, ad)
    L = _lib.lib()
    N, C = a3.shape[:2]
    P = a3[0, 0].numel()
    tiles = L.x3d_ew_tiles(P)
    g = g if g is not None else _f(a3.shape, a3)
    partial = _f((N, C, tiles, 2), a3)
    partial_d = _f((N, C, tiles, 2), a3) if ad is not None else None
    check(L.x3d_bn_add_relu_bwd(ptr(dout), ptr(out), ptr(a3), ptr(ad), ptr(g), ptr(partial), ptr(partial_d), N, C, P,
                                _lib.stream()))
    return g, partial, partial_d


def bn_relu_pool_fwd(a5, c5, per_frame=False):
    """per_frame=False: pooled [N, C] (task 'class'); True: [N, C, T] (task 'loc', pooling over H, W only)."""
    _need_cuda(a5, c5)
    L = _lib.lib()
    N, C = a5.shape[:2]
    P = a5[0, 0].numel()
    segs = a5.shape[2] if per_frame else 1
    pooled = _f((N, C, segs) if per_frame else (N, C), a5)
    check(L.x3d_bn_relu_pool_fwd(ptr(a5), ptr(c5), ptr(pooled), N, C, P, segs, _lib.stream()))
    return pooled


def bn_relu_pool_bwd(a5, c5, dpooled, g=None):
    _need_cuda(a5, c5, dpooled)
    L = _lib.lib()
    N, C = a5.shape[:2]
    P = a5[0, 0].numel()
    g = g if g is not None else _f(a5.shape, a5)
    partial = _f((N, C, L.x3d_ew_tiles(P), 2), a5)
    segs = dpooled.shape[2] if dpooled.dim() == 3 else 1
    check(L.x3d_bn_relu_pool_bwd(ptr(a5), ptr(c5), ptr(dpooled), ptr(g), ptr(partial), N, C, P, segs, _lib.stream()))
    return g, partial


# ----------------------------------------------------------------------------- head (fc1 / dropout / fc2 / CE)
def head_rng_state(dev, seed=None):
    """Device {seed, draw counter} of the head's dropout (two uint64 kept in an int64 tensor)."""
    if seed is None:
        seed = torch.initial_seed()
    return torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)


def head_fwd(pooled, w1, w2, b2, p_drop=0.0, rng=None):
    """pooled [R, K] -> (hd [R, J] kept for the backward, logits [R, C])."""
    _need_cuda(pooled, w1, w2, b2)
    R, K = pooled.shape
    J, C = w1.shape[0], w2.shape[0]
    hd, logits = _f((R, J), pooled), _f((R, C), pooled)
    check(_lib.lib().x3d_head_fwd(ptr(pooled), ptr(w1), ptr(w2), ptr(b2), ptr(hd), ptr(logits), R, K, J, C, float(p_drop),
                                  ptr(rng) if p_drop > 0 else None, _lib.stream()))
    return hd, logits


def head_advance_rng(rng):
    d = _f((1,), rng.new_zeros(1, dtype=torch.float32))
    check(_lib.lib().x3d_head_advance_rng(ptr(rng), ptr(d), _lib.stream()))


def head_ce(logits, labels, rng=None):
    """Mean cross entropy over the rows of logits [R, C] vs labels [R] (int64): (loss [1], dlogits [R, C])."""
    _need_cuda(logits)
    R, C = logits.shape
    if labels.dtype != torch.int64 or not labels.is_contiguous() or labels.numel() != R:
        raise ValueError("head_ce: labels must be contiguous int64 [R]")
    loss, dlog, sc = _f((1,), logits), _f((R, C), logits), _f((R,), logits)
    check(_lib.lib().x3d_head_ce(ptr(logits), ptr(labels), ptr(loss), ptr(dlog), ptr(sc), R, C, ptr(rng), _lib.stream()))
    return loss, dlog


def head_bwd(dlogits, hd, pooled, w1, w2, p_drop=0.0, outs=None):
    """(dpooled [R, K], dW1 [J, K], dW2 [C, J], db2 [C]); outs = (dW1, dW2, db2) writes into existing storage."""
    _need_cuda(dlogits, hd, pooled, w1, w2)
    L = _lib.lib()
    R, K = pooled.shape
    J, C = w1.shape[0], w2.shape[0]
    dw1, dw2, db2 = outs if outs is not None else (_f((J, K), pooled), _f((C, J), pooled), _f((C,), pooled))
    dpooled = _f((R, K), pooled)
    sc = _f((L.x3d_head_scratch_floats(R, K, J, C),), pooled)
    check(L.x3d_head_bwd(ptr(dlogits), ptr(hd), ptr(pooled), ptr(w1), ptr(w2), ptr(dw1), ptr(dw2), ptr(db2), ptr(dpooled),
                         ptr(sc), R, K, J, C, float(p_drop), _lib.stream()))
    return dpooled, dw1, dw2, db2


def loc_losses(logits, labels, grad_scale=0.5):
    """Charades localisation losses on per-frame logits [B, C, T] vs float labels [B, C, TL]:
    (losses [2] = (cls_loss, loc_loss), dlogits [B, C, T] of (cls + loc) * grad_scale)."""
    _need_cuda(logits, labels)
    B, C, T = logits.shape
    TL = labels.shape[2]
    losses, dlog, sc = _f((2,), logits), _f((B, C, T), logits), _f((2 * B * C,), logits)
    check(_lib.lib().x3d_loc_losses(ptr(logits), ptr(labels), ptr(losses), ptr(dlog), ptr(sc), B, C, T, TL, float(grad_scale),
                                    _lib.stream()))
    return losses, dlog


def bn_rowstats(x, g=None):
    _need_cuda(x, g)
    L = _lib.lib()
    N, C = x.shape[:2]
    P = x[0, 0].numel()
    partial = _f((N, C, L.x3d_ew_tiles(P), 2), x)
    check(L.x3d_bn_rowstats(ptr(x), ptr(g), ptr(partial), N, C, P, _lib.stream()))
    return partial


def bn_affine(x, coef, g=None):
    _need_cuda(x, coef, g)
    N, C = x.shape[:2]
    P = x[0, 0].numel()
    out = _f(x.shape, x)
    check(_lib.lib().x3d_bn_affine(ptr(x), ptr(g), ptr(coef), ptr(out), N, C, P, coef.shape[-1], _lib.stream()))
    return out


def grad_accumulate(acc, g, scale, first):
    _need_cuda(acc, g)
    check(_lib.lib().x3d_grad_accumulate(ptr(acc), ptr(g), g.numel(), scale, 1 if first else 0, _lib.stream()))


def sgd_fused(w, g, m, lr, momentum=0.9, weight_decay=5e-5, grad_scale=1.0, first=False):
    _need_cuda(w, g, m)
    check(_lib.lib().x3d_sgd_fused(ptr(w), ptr(g), ptr(m), w.numel(), lr, momentum, weight_decay, grad_scale,
                                   1 if first else 0, _lib.stream()))
