"""X3D (S / M / XL, plus 'L' = XL depth at M width) for MI355X -- drop-in for the reference's
``x3d.py`` module API, executed by hand-written HIP kernels (libx3dhip.so).

Same public surface as /root/reference/x3d.py:
  generate_model(x3d_version, **kw)              x3d.py:366-368
  ResNet(block, layers, block_inplanes, ...)     x3d.py:174-186   (same keyword arguments)
    .forward(x[B,3,T,H,W]) -> [B,n_classes,1]    x3d.py:316-345
    .update_bn_splits_long_cycle(scale) -> int   x3d.py:298-303
    .aggregate_sub_bn_stats() -> int             x3d.py:306-313
    .replace_logits(n_classes)                   x3d.py:294-295
  Bottleneck, SubBatchNorm3d, Swish, SwishEfficient, conv3x3x3, conv1x1x1,
  get_inplanes, get_blocks
and the same ``state_dict()`` layout (820 entries for S/M: ``<m>.weight/.bias``,
``<m>.bn.running_*``, ``<m>.split_bn.running_*``, ``num_batches_tracked``), so checkpoints in
the reference's format load unchanged.

What differs is *how* forward/backward run: the modules below only own parameters and
buffers; ``ResNet.forward`` hands them to ``x3dhip.engine`` which schedules fused HIP
kernels over NCTHW fp32 tensors.  There is no eager-PyTorch or CPU fallback: a CPU input or a
missing libx3dhip.so raises.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from x3dhip import engine as _engine
from x3dhip import ops as _ops
from x3dhip import _lib as _hiplib


# ----------------------------------------------------------------------------------------
# parameter containers
# ----------------------------------------------------------------------------------------
class _Conv3dParams(nn.Module):
    """Weights (and optional bias) of a Conv3d; the arithmetic happens in the HIP kernels."""

    def __init__(self, cin, cout, kernel, stride=(1, 1, 1), padding=(0, 0, 0), groups=1, bias=False):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride, self.padding, self.groups = tuple(kernel), tuple(stride), tuple(padding), groups
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, *kernel))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # reference init, x3d.py:246-250: kaiming normal, fan_out, relu gain, on every Conv3d;
        # conv biases keep nn.Conv3d's default U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        fan_out = self.out_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        with torch.no_grad():
            self.weight.normal_(0.0, math.sqrt(2.0 / fan_out))
            if self.bias is not None:
                fan_in = (self.in_channels // self.groups) * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
                b = 1.0 / math.sqrt(fan_in)
                self.bias.uniform_(-b, b)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%s, groups=%d" % (self.in_channels, self.out_channels,
                                                                 self.kernel_size, self.stride, self.groups)


def conv3x3x3(in_planes, out_planes, stride=1):
    """Channelwise 3x3x3, stride (1,s,s), pad 1 (x3d.py:87-95)."""
    return _Conv3dParams(in_planes, out_planes, (3, 3, 3), (1, stride, stride), (1, 1, 1), groups=in_planes)


def conv1x1x1(in_planes, out_planes, stride=1):
    """Pointwise, stride (1,s,s) (x3d.py:98-103)."""
    return _Conv3dParams(in_planes, out_planes, (1, 1, 1), (1, stride, stride))


class _BNStats(nn.Module):
    """running_mean / running_var / num_batches_tracked of a non-affine nn.BatchNorm3d."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.track_running_stats = True
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class SubBatchNorm3d(nn.Module):
    """Split batch norm (x3d.py:9-58): in training, sample n is normalised with the
    statistics of split n % num_splits; one shared affine; ``bn`` holds the aggregated
    statistics used in eval."""

    def __init__(self, num_splits, **args):
        super().__init__()
        self.num_splits = num_splits
        self.num_features = args["num_features"]
        self.affine = args.get("affine", True)
        if self.affine:
            self.weight = nn.Parameter(torch.ones(self.num_features))
            self.bias = nn.Parameter(torch.zeros(self.num_features))
        self.bn = _BNStats(self.num_features)
        self.split_bn = _BNStats(self.num_features * self.num_splits)

    def aggregate_stats(self):
        """Fold the per-split running statistics into ``bn`` (x3d.py:27-45)."""
        S = self.num_splits
        mu = self.split_bn.running_mean.view(S, -1)
        var = self.split_bn.running_var.view(S, -1)
        mean = mu.sum(0) / S
        self.bn.running_mean.data = mean.detach()
        self.bn.running_var.data = (var.sum(0) / S + ((mu - mean) ** 2).sum(0) / S).detach()

    def forward(self, x):
        """Stand-alone use (x3d.py:47-58).  Inside ResNet the statistics ride in the producing conv's epilogue and the
        normalisation in the consumer's load; called as a module of its own it is two row passes over the HIP kernels."""
        if not x.is_cuda:
            raise _hiplib.X3DHipError("SubBatchNorm3d.forward needs a CUDA(HIP) tensor (no CPU path)")
        return _SubBNFunction.apply(self, x.contiguous().float(), self.weight, self.bias)


class _SubBNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, weight, bias):
        N, C = x.shape[:2]
        P = x[0, 0].numel()
        S = mod.num_splits
        if mod.training:
            if N % S != 0:
                raise ValueError("batch size %d is not divisible by num_splits %d (x3d.py:50)" % (N, S))
            coef, save, _ = _ops.bn_fwd_finalize(_ops.bn_rowstats(x), S, P, weight.data, bias.data,
                                                 mod.split_bn.running_mean, mod.split_bn.running_var,
                                                 _engine.BN_MOMENTUM, _engine.BN_EPS)
            mod.split_bn.num_batches_tracked += 1
            ctx.save_for_backward(x, save, weight)
            ctx.S, ctx.P = S, P
        else:
            coef = _ops.bn_eval_coef(mod.bn.running_mean, mod.bn.running_var, weight.data, bias.data, N, _engine.BN_EPS)
            ctx.save_for_backward(coef)
            ctx.S = 0
        return _ops.bn_affine(x, coef)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        if ctx.S == 0:                  # eval: y = c0 * x + c1 with constant statistics
            (coef,) = ctx.saved_tensors
            c0 = coef[..., 0].contiguous()
            dx = g * c0.view(c0.shape[0], c0.shape[1], *([1] * (g.dim() - 2)))
            return None, dx, None, None
        x, save, weight = ctx.saved_tensors
        cb, dgamma, dbeta = _ops.bn_bwd_finalize(_ops.bn_rowstats(x, g), ctx.S, ctx.P, weight.data, save)
        return None, _ops.bn_affine(x, cb, g), dgamma, dbeta


class _BlockPacks:
    """Packed weights of one block, built on demand (ResNet keeps one batched pack for the whole model instead)."""

    def get(self, w, transposed=False):
        return _ops.pw_pack(w.data.view(w.shape[0], -1), transposed=transposed)


class _BlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk, x, *params):
        S = blk.bn1.num_splits
        if blk.training and x.shape[0] % S != 0:
            raise ValueError("batch size %d is not divisible by num_splits %d (x3d.py:50)" % (x.shape[0], S))
        tctx = _engine.TrunkContext()
        out, _ = _engine._block_forward(blk, x, None, S, blk.training, tctx if blk.training else None, _BlockPacks())
        if blk.training:
            for m in (blk.bn1, blk.bn2, blk.bn3) + ((blk.downsample[1],) if blk.downsample is not None else ()):
                m.split_bn.num_batches_tracked += 1
        ctx.blk, ctx.tctx, ctx.params = blk, tctx, params
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.blk.training:
            raise RuntimeError("Bottleneck.forward in eval mode is inference only")
        sink = _engine._GradSink(False)
        dprev, _ = _engine._block_backward(ctx.tctx.blocks[0], dout.contiguous().float(), sink)
        sink.flush()
        ctx.tctx = None
        return (None, dprev) + tuple(sink.written[p].view(p.shape) for p in ctx.params)


class SwishEfficient(torch.autograd.Function):
    """x * sigmoid(x) with the reference's hand-written backward (x3d.py:71-84).  Inside the
    network the op is fused into conv3's load (forward) and conv3's data-gradient epilogue
    (backward); this standalone form exists for API parity."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x * torch.sigmoid(x)

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        s = torch.sigmoid(x)
        return grad_output * (s * (1 + x * (1 - s)))


class Swish(nn.Module):
    def forward(self, x):
        return SwishEfficient.apply(x)


class Bottleneck(nn.Module):
    """Inverted bottleneck (x3d.py:106-171): 1x1x1 -> BN/ReLU -> channelwise 3x3x3 -> BN ->
    [SE on even index] -> Swish -> 1x1x1 -> BN -> + residual -> ReLU."""

    def __init__(self, in_planes, planes, stride=1, downsample=None, index=0, base_bn_splits=8):
        super().__init__()
        self.index = index
        self.base_bn_splits = base_bn_splits
        self.stride = stride
        self.has_se = index % 2 == 0
        mid, out = planes
        self.conv1 = conv1x1x1(in_planes, mid)
        self.bn1 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=mid, affine=True)
        self.conv2 = conv3x3x3(mid, mid, stride)
        self.bn2 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=mid, affine=True)
        self.conv3 = conv1x1x1(mid, out)
        self.bn3 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=out, affine=True)
        self.swish = Swish()
        if self.has_se:
            width = self.round_width(mid)
            self.fc1 = _Conv3dParams(mid, width, (1, 1, 1), bias=True)
            self.fc2 = _Conv3dParams(width, mid, (1, 1, 1), bias=True)
        self.downsample = downsample

    @staticmethod
    def round_width(width, multiplier=0.0625, min_width=8, divisor=8):
        """SE bottleneck width (x3d.py:129-140)."""
        if not multiplier:
            return width
        width *= multiplier
        min_width = min_width or divisor
        width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
        if width_out < 0.9 * width:
            width_out += divisor
        return int(width_out)

    def forward(self, x):
        """Stand-alone use (x3d.py:143-171): the same fused HIP schedule ResNet.forward runs per block."""
        if not x.is_cuda:
            raise _hiplib.X3DHipError("Bottleneck.forward needs a CUDA(HIP) tensor (no CPU path)")
        return _BlockFunction.apply(self, x.contiguous().float(), *[p for p in self.parameters()])


_PLANES = {"S": [(54, 24), (108, 48), (216, 96), (432, 192)],
           "M": [(54, 24), (108, 48), (216, 96), (432, 192)],
           "L": [(54, 24), (108, 48), (216, 96), (432, 192)],    # BASELINE cfg 5: XL depth, M width
           "XL": [(72, 32), (162, 72), (306, 136), (630, 280)]}
_BLOCKS = {"S": [3, 5, 11, 7], "M": [3, 5, 11, 7], "L": [5, 10, 25, 15], "XL": [5, 10, 25, 15]}


def get_inplanes(version):
    return _PLANES[version]


def get_blocks(version):
    return _BLOCKS[version]


class ResNet(nn.Module):

    def __init__(self, block, layers, block_inplanes, n_input_channels=3, shortcut_type='B', widen_factor=1.0,
                 dropout=0.5, n_classes=400, base_bn_splits=8, task='class', act_dtype=torch.float32):
        super().__init__()
        if shortcut_type != 'B':
            raise NotImplementedError("only shortcut_type 'B' (conv + BN downsample) is built; "
                                      "type 'A' is dead code in the reference (x3d.py:252-261)")
        if task not in ('class', 'loc'):
            raise ValueError("task must be 'class' or 'loc'")
        block_inplanes = [(int(a * widen_factor), int(b * widen_factor)) for a, b in block_inplanes]
        self.base_bn_splits = base_bn_splits
        self.task = task
        # storage type of the wide tensors inside the bottlenecks (x3dhip.engine.wide_dtype): torch.bfloat16 selects the
        # mixed-storage mode (bf16 storage / fp32 accumulate); not in the reference, whose tensors are all fp32
        self.act_dtype = act_dtype
        self.in_planes = block_inplanes[0][1]
        self.index = 0
        self._pending_tracked = 0
        self._bn_version = 0      # bumped when split_bn buffers are re-created (graph caches key on it)

        self.conv1_s = _Conv3dParams(n_input_channels, self.in_planes, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.conv1_t = _Conv3dParams(self.in_planes, self.in_planes, (5, 1, 1), (1, 1, 1), (2, 0, 0),
                                     groups=self.in_planes)
        self.bn1 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=self.in_planes, affine=True)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block, block_inplanes[0], layers[0], stride=2)
        self.layer2 = self._make_layer(block, block_inplanes[1], layers[1], stride=2)
        self.layer3 = self._make_layer(block, block_inplanes[2], layers[2], stride=2)
        self.layer4 = self._make_layer(block, block_inplanes[3], layers[3], stride=2)
        self.conv5 = _Conv3dParams(block_inplanes[3][1], block_inplanes[3][0], (1, 1, 1))
        self.bn5 = SubBatchNorm3d(num_splits=base_bn_splits, num_features=block_inplanes[3][0], affine=True)
        self.fc1 = _Conv3dParams(block_inplanes[3][0], 2048, (1, 1, 1))
        self.fc2 = nn.Linear(2048, n_classes)
        self.dropout = nn.Dropout(dropout)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.in_planes != planes[1]:
            downsample = nn.Sequential(
                conv1x1x1(self.in_planes, planes[1], stride),
                SubBatchNorm3d(num_splits=self.base_bn_splits, num_features=planes[1], affine=True))
        mods = [block(in_planes=self.in_planes, planes=planes, stride=stride, downsample=downsample,
                      index=0, base_bn_splits=self.base_bn_splits)]
        self.in_planes = planes[1]
        for i in range(1, blocks):
            mods.append(block(self.in_planes, planes, index=i, base_bn_splits=self.base_bn_splits))
        return nn.Sequential(*mods)

    # ------------------------------------------------------------------ reference methods
    def replace_logits(self, n_classes):
        dev = self.fc2.weight.device
        self.fc2 = nn.Linear(2048, n_classes).to(dev)

    def update_bn_splits_long_cycle(self, long_cycle_bn_scale):
        """Re-create every split_bn for num_splits = base * scale (fresh running stats, as the
        reference does at each long-cycle switch, x3d.py:298-303)."""
        self._flush_tracked()
        n = self.base_bn_splits * long_cycle_bn_scale
        for m in self.modules():
            if isinstance(m, SubBatchNorm3d):
                m.num_splits = n
                m.split_bn = _BNStats(m.num_features * n).to(m.weight.device)
        self._bn_version += 1
        return n

    def aggregate_sub_bn_stats(self):
        count = 0
        for m in self.modules():
            if isinstance(m, SubBatchNorm3d):
                m.aggregate_stats()
                count += 1
        return count

    # ------------------------------------------------------------------ bookkeeping
    def _flush_tracked(self):
        """num_batches_tracked is advanced lazily (one host counter per model instead of 84
        one-element kernels per step); flushed whenever state is observed or re-shaped."""
        if self._pending_tracked:
            k = self._pending_tracked
            self._pending_tracked = 0
            for m in self.modules():
                if isinstance(m, SubBatchNorm3d):
                    m.split_bn.num_batches_tracked += k

    def state_dict(self, *args, **kwargs):
        self._flush_tracked()
        return super().state_dict(*args, **kwargs)

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not x.is_cuda:
            raise _hiplib.X3DHipError("x3d.ResNet.forward needs a CUDA(HIP) tensor: the network runs on "
                                      "hand-written MI355X kernels and has no CPU path")
        _hiplib.lib()   # raises when libx3dhip.so is missing
        x = x.contiguous().float()
        if self.training and torch.is_grad_enabled():
            pooled = _engine.TrunkFunction.apply(self, x, *_engine.trunk_parameters(self))
            self._pending_tracked += 1
        else:
            with torch.no_grad():
                pooled = _engine.trunk_forward(self, x, self.training, None)
            if self.training:
                self._pending_tracked += 1
        # head (x3d.py:333-343): 1x1x1 conv on the pooled vector(s) == linear on rows; HIP kernels (csrc/head.hip)
        if self.task == 'loc':
            # pooled [B, C5, T]: per-frame rows [B*T, C5] -> logits [B*T, n_classes] -> [B, n_classes, T]
            B, C5, T = pooled.shape
            rows = pooled.permute(0, 2, 1).reshape(B * T, C5).contiguous()
            logits = _HeadFunction.apply(self, rows, self.fc1.weight, self.fc2.weight, self.fc2.bias)
            return logits.view(B, T, -1).permute(0, 2, 1)
        logits = _HeadFunction.apply(self, pooled, self.fc1.weight, self.fc2.weight, self.fc2.bias)
        return logits.unsqueeze(2)

    def _head_rng(self, dev):
        """Device {seed, draw counter} of the head's dropout (created on first use from torch's seed)."""
        st = getattr(self, "_head_rng_state", None)
        if st is None or st.device != dev:
            st = _ops.head_rng_state(dev)
            self._head_rng_state = st
        return st


class _HeadFunction(torch.autograd.Function):
    """fc1 -> ReLU -> Dropout -> fc2 (x3d.py:333-339) on the pooled rows, through libx3dhip's head kernels."""

    @staticmethod
    def forward(ctx, model, pooled, w1, w2, b2):
        p = float(model.dropout.p) if model.training else 0.0
        rng = model._head_rng(pooled.device) if p > 0 else None
        w1v = w1.view(w1.shape[0], -1)
        hd, logits = _ops.head_fwd(pooled, w1v, w2, b2, p, rng)
        if rng is not None:
            _ops.head_advance_rng(rng)          # the next forward draws a fresh dropout mask
        ctx.save_for_backward(hd, pooled, w1, w2)
        ctx.p = p
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        hd, pooled, w1, w2 = ctx.saved_tensors
        dpooled, dw1, dw2, db2 = _ops.head_bwd(dlogits.contiguous().float(), hd, pooled, w1.view(w1.shape[0], -1), w2, ctx.p)
        return None, dpooled, dw1.view_as(w1), dw2, db2


def generate_model(x3d_version, **kwargs):
    return ResNet(Bottleneck, get_blocks(x3d_version), get_inplanes(x3d_version), **kwargs)
