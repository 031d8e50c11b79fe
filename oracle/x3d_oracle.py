"""ORACLE (test infrastructure only) -- CPU restatement of the X3D forward/backward.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``x3d-multigrid_amd/x3d.py``) runs hand-written HIP kernels and
raises when the HIP library is missing; it never routes through this file.

What it restates (reference = /root/reference, read-only):
  * network topology and widths            x3d.py:174-291, 352-363
  * stem 1x3x3 dense s(1,2,2) + 5x1x1 dw   x3d.py:196-208, 317-320
  * bottleneck block                        x3d.py:143-171
  * split batch-norm (interleaved splits)   x3d.py:9-58
  * swish                                   x3d.py:61-84
  * squeeze-excitation + round_width        x3d.py:120-140, 153-159
  * head                                    x3d.py:231-244, 327-339
  * sub-BN stat aggregation                 x3d.py:27-45, 306-313
  * train step loss                         train_x3d_kinetics_multigrid.py:189,245,259

The arithmetic itself lives in a third-party dependency of the reference that is
not vendored under /root/reference: PyTorch (pinned by the reference at 1.7.0 +
PR #40801, README.md:30-34).  Here the same stock ATen CPU ops of the torch wheel in
this image (2.10.0) are used: conv3d, mean/var reductions, sigmoid.  The
restatement is *functional* (explicit tensors in, tensors out, BN written out as
mean/var arithmetic) rather than an nn.Module tree.

PINNING: the functions below are pinned by tests/golden/*.npz, produced by
tests/golden/make_golden.py which imports the reference's own x3d.py in the build
container (see tests/test_oracle_golden.py).  The reference repo has no tests or
golden vectors of its own.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm3d default used at x3d.py:23,25
BN_MOMENTUM = 0.1    # idem

# x3d.py:352-363 (S and M are identical networks; 'L' = XL depth with M widths,
# SURVEY.md 3.6 -- not a reference key, used only by BASELINE config 5)
_WIDTHS = {
    "S": [(54, 24), (108, 48), (216, 96), (432, 192)],
    "M": [(54, 24), (108, 48), (216, 96), (432, 192)],
    "L": [(54, 24), (108, 48), (216, 96), (432, 192)],
    "XL": [(72, 32), (162, 72), (306, 136), (630, 280)],
}
_DEPTHS = {"S": [3, 5, 11, 7], "M": [3, 5, 11, 7], "L": [5, 10, 25, 15], "XL": [5, 10, 25, 15]}


def se_width(c, multiplier=0.0625, min_width=8, divisor=8):
    """x3d.py:129-140."""
    w = c * multiplier
    out = max(min_width, int(w + divisor / 2) // divisor * divisor)
    if out < 0.9 * w:
        out += divisor
    return int(out)


def block_table(version):
    """[(prefix, cin, cmid, cout, stride, has_se, has_downsample)] in forward order
    (x3d.py:263-291: SE on even block index, index restarts per stage; first block
    of each stage has stride 2 and a 1x1x1 stride-2 downsample + SubBN)."""
    rows = []
    cin = _WIDTHS[version][0][1]
    for s, ((cm, co), n) in enumerate(zip(_WIDTHS[version], _DEPTHS[version])):
        for b in range(n):
            first = b == 0
            rows.append(("layer%d.%d" % (s + 1, b), cin, cm, co, 2 if first else 1,
                         b % 2 == 0, first))
            cin = co
    return rows


def _bn_entries(sd, prefix, c, splits):
    sd[prefix + ".weight"] = torch.ones(c)
    sd[prefix + ".bias"] = torch.zeros(c)
    sd[prefix + ".bn.running_mean"] = torch.zeros(c)
    sd[prefix + ".bn.running_var"] = torch.ones(c)
    sd[prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    sd[prefix + ".split_bn.running_mean"] = torch.zeros(c * splits)
    sd[prefix + ".split_bn.running_var"] = torch.ones(c * splits)
    sd[prefix + ".split_bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def state_template(version="M", n_classes=400, num_splits=1, n_input_channels=3):
    """Zero/one-filled state dict with the reference's key order and shapes
    (x3d.py registration order; 820 entries for S/M)."""
    sd = OrderedDict()
    c0 = _WIDTHS[version][0][1]
    sd["conv1_s.weight"] = torch.zeros(c0, n_input_channels, 1, 3, 3)
    sd["conv1_t.weight"] = torch.zeros(c0, 1, 5, 1, 1)
    _bn_entries(sd, "bn1", c0, num_splits)
    for (p, cin, cm, co, stride, se, ds) in block_table(version):
        sd[p + ".conv1.weight"] = torch.zeros(cm, cin, 1, 1, 1)
        _bn_entries(sd, p + ".bn1", cm, num_splits)
        sd[p + ".conv2.weight"] = torch.zeros(cm, 1, 3, 3, 3)
        _bn_entries(sd, p + ".bn2", cm, num_splits)
        sd[p + ".conv3.weight"] = torch.zeros(co, cm, 1, 1, 1)
        _bn_entries(sd, p + ".bn3", co, num_splits)
        if se:
            w = se_width(cm)
            sd[p + ".fc1.weight"] = torch.zeros(w, cm, 1, 1, 1)
            sd[p + ".fc1.bias"] = torch.zeros(w)
            sd[p + ".fc2.weight"] = torch.zeros(cm, w, 1, 1, 1)
            sd[p + ".fc2.bias"] = torch.zeros(cm)
        if ds:
            sd[p + ".downsample.0.weight"] = torch.zeros(co, cin, 1, 1, 1)
            _bn_entries(sd, p + ".downsample.1", co, num_splits)
    c4m, c4o = _WIDTHS[version][3]
    sd["conv5.weight"] = torch.zeros(c4m, c4o, 1, 1, 1)
    _bn_entries(sd, "bn5", c4m, num_splits)
    sd["fc1.weight"] = torch.zeros(2048, c4m, 1, 1, 1)
    sd["fc2.weight"] = torch.zeros(n_classes, 2048)
    sd["fc2.bias"] = torch.zeros(n_classes)
    return sd


def is_parameter(name):
    """True for entries that are nn.Parameters in the reference (get gradients)."""
    leaf = name.rsplit(".", 1)[-1]
    return leaf in ("weight", "bias")


def split_bn(x, sd, prefix, S, training, new_stats):
    """x3d.py:47-58.  Training: sample n belongs to split n % S (the view
    [N/S, S*C, ...] interleaves); per-(split, channel) biased variance normalises,
    running stats (layout j*C + c) take the unbiased variance with momentum 0.1.
    Eval: the aggregated ``bn`` running stats."""
    N, C = x.shape[0], x.shape[1]
    if training:
        xs = x.reshape(N // S, S, C, *x.shape[2:])
        red = (0, 3, 4, 5)
        mean = xs.mean(dim=red, keepdim=True)
        var = ((xs - mean) ** 2).mean(dim=red, keepdim=True)
        y = ((xs - mean) / torch.sqrt(var + BN_EPS)).reshape(x.shape)
        if new_stats is not None:
            cnt = xs.numel() // (S * C)
            m = mean.detach().reshape(S * C)
            v = var.detach().reshape(S * C) * (cnt / max(cnt - 1, 1))
            rm, rv = sd[prefix + ".split_bn.running_mean"], sd[prefix + ".split_bn.running_var"]
            new_stats[prefix + ".split_bn.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * m
            new_stats[prefix + ".split_bn.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * v
            new_stats[prefix + ".split_bn.num_batches_tracked"] = \
                sd[prefix + ".split_bn.num_batches_tracked"] + 1
    else:
        rm = sd[prefix + ".bn.running_mean"].view(1, C, 1, 1, 1)
        rv = sd[prefix + ".bn.running_var"].view(1, C, 1, 1, 1)
        y = (x - rm) / torch.sqrt(rv + BN_EPS)
    y = y * sd[prefix + ".weight"].view(1, C, 1, 1, 1)
    y = y + sd[prefix + ".bias"].view(1, C, 1, 1, 1)
    return y


def swish(x):
    """x3d.py:75-84 (forward x*sigmoid(x); the hand-written backward there is the
    analytic derivative, which autograd reproduces)."""
    return x * torch.sigmoid(x)


class _StoreBf16(torch.autograd.Function):
    """y = bf16(x) widened again (round to nearest even); the gradient passes unchanged."""
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _StoreGradBf16(torch.autograd.Function):
    """Identity whose incoming gradient is rounded to bf16."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


# Mixed-storage mode of the product (NOT in the reference, whose tensors are all fp32; include/x3dhip.h X3D_MX_*,
# BASELINE config 5 "bf16 storage / fp32 accumulate"): when True, the four wide tensors of every bottleneck are rounded
# to bf16 where the product stores them -- conv1's and conv2's raw outputs in the forward pass, and the gradients with
# respect to the two pre-activations (bn1's output; bn2's output after the SE gate) in the backward pass.  Everything
# else, including all arithmetic, is the restatement of x3d.py above and below.  Tests set this flag around a call.
BF16_WIDE = False


def bottleneck(x, sd, p, stride, has_se, has_ds, S, training, new_stats):
    """x3d.py:143-171."""
    cm = sd[p + ".conv2.weight"].shape[0]
    q = BF16_WIDE
    out = F.conv3d(x, sd[p + ".conv1.weight"])
    if q:
        out = _StoreBf16.apply(out)
    out = split_bn(out, sd, p + ".bn1", S, training, new_stats)
    if q:
        out = _StoreGradBf16.apply(out)
    out = torch.relu(out)
    out = F.conv3d(out, sd[p + ".conv2.weight"], stride=(1, stride, stride), padding=1, groups=cm)
    if q:
        out = _StoreBf16.apply(out)
    out = split_bn(out, sd, p + ".bn2", S, training, new_stats)
    if has_se:
        g = out.mean(dim=(2, 3, 4), keepdim=True)
        g = torch.relu(F.conv3d(g, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"]))
        g = torch.sigmoid(F.conv3d(g, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"]))
        out = out * g
    if q:
        out = _StoreGradBf16.apply(out)
    out = swish(out)
    out = F.conv3d(out, sd[p + ".conv3.weight"])
    out = split_bn(out, sd, p + ".bn3", S, training, new_stats)
    res = x
    if has_ds:
        res = F.conv3d(x, sd[p + ".downsample.0.weight"], stride=(1, stride, stride))
        res = split_bn(res, sd, p + ".downsample.1", S, training, new_stats)
    return torch.relu(out + res)


def trunk(x, sd, version, S, training, new_stats=None, taps=None):
    """Stem + the four stages + conv5/bn5/relu (x3d.py:317-329)."""
    c0 = sd["conv1_s.weight"].shape[0]
    x = F.conv3d(x, sd["conv1_s.weight"], stride=(1, 2, 2), padding=(0, 1, 1))
    x = F.conv3d(x, sd["conv1_t.weight"], padding=(2, 0, 0), groups=c0)
    x = torch.relu(split_bn(x, sd, "bn1", S, training, new_stats))
    if taps is not None:
        taps["stem"] = x
    for (p, cin, cm, co, stride, se, ds) in block_table(version):
        x = bottleneck(x, sd, p, stride, se, ds, S, training, new_stats)
        if taps is not None:
            taps[p] = x
    x = F.conv3d(x, sd["conv5.weight"])
    x = torch.relu(split_bn(x, sd, "bn5", S, training, new_stats))
    return x


def forward(x, sd, version="M", num_splits=1, training=True, new_stats=None, taps=None,
            dropout_p=0.0, task="class"):
    """Logits [B, n_classes, 1] (x3d.py:316-345, task='class'); 'loc' gives [B, n_classes, T]."""
    x = trunk(x, sd, version, num_splits, training, new_stats, taps)
    if task == "class":
        x = x.mean(dim=(2, 3, 4), keepdim=True)
    else:
        x = x.mean(dim=(3, 4), keepdim=True)
    x = torch.relu(F.conv3d(x, sd["fc1.weight"]))
    if task == "class":
        x = x.flatten(1)
        if dropout_p > 0 and training:
            x = F.dropout(x, dropout_p, True)
        return F.linear(x, sd["fc2.weight"], sd["fc2.bias"]).unsqueeze(2)
    x = x.squeeze(4).squeeze(3).permute(0, 2, 1)
    if dropout_p > 0 and training:
        x = F.dropout(x, dropout_p, True)
    return F.linear(x, sd["fc2.weight"], sd["fc2.bias"]).permute(0, 2, 1)


def loss_fn(logits, labels):
    """CrossEntropyLoss on logits[B,C,1] vs labels[B,1], mean (train...:189,245,259)."""
    return F.cross_entropy(logits, labels)


def train_step_grads(x, labels, sd, version="M", num_splits=1):
    """One forward + loss + backward.  Returns (logits, loss, grads{name}, new_stats)."""
    leaf = OrderedDict()
    for k, v in sd.items():
        if is_parameter(k):
            leaf[k] = v.detach().clone().requires_grad_(True)
        else:
            leaf[k] = v
    new_stats = {}
    logits = forward(x, leaf, version, num_splits, True, new_stats)
    loss = loss_fn(logits, labels)
    names = [k for k in leaf if is_parameter(k)]
    gs = torch.autograd.grad(loss, [leaf[k] for k in names])
    return logits.detach(), loss.detach(), OrderedDict(zip(names, gs)), new_stats


def aggregate_sub_bn(sd, num_splits):
    """x3d.py:27-45: bn.running_mean = mean_j mu_j; bn.running_var = mean_j var_j +
    mean_j (mu_j - mean)^2.  Returns {name: tensor} for every '.bn.running_*'."""
    out = {}
    for k in sd:
        if k.endswith(".split_bn.running_mean"):
            p = k[: -len(".split_bn.running_mean")]
            mu = sd[k].view(num_splits, -1)
            var = sd[p + ".split_bn.running_var"].view(num_splits, -1)
            m = mu.sum(0) / num_splits
            out[p + ".bn.running_mean"] = m
            out[p + ".bn.running_var"] = var.sum(0) / num_splits + ((mu - m) ** 2).sum(0) / num_splits
    return out


def sgd_step(params, grads, moms, lr, momentum=0.9, weight_decay=5e-5):
    """torch.optim.SGD semantics used at train...:183 (wd on every parameter,
    first step initialises the momentum buffer with the gradient)."""
    new_p, new_m = {}, {}
    for k in params:
        g = grads[k] + weight_decay * params[k]
        m = g.clone() if moms.get(k) is None else momentum * moms[k] + g
        new_m[k] = m
        new_p[k] = params[k] - lr * m
    return new_p, new_m


# ---------------------------------------------------------------------------
# Per-op oracles (float64) used by the kernel-level parity tests.
# ---------------------------------------------------------------------------

def dw333(x, w, stride):
    c = x.shape[1]
    return F.conv3d(x, w, stride=(1, stride, stride), padding=1, groups=c)


def dw5t(x, w):
    c = x.shape[1]
    return F.conv3d(x, w, padding=(2, 0, 0), groups=c)


def pw(x, w, stride=1):
    return F.conv3d(x, w, stride=(1, stride, stride))


def stem133(x, w):
    return F.conv3d(x, w, stride=(1, 2, 2), padding=(0, 1, 1))


def out_hw(h, stride):
    return (h - 1) // stride + 1 if stride > 1 else h
