"""Experiment (not a test): how much do split-bf16 pointwise contractions perturb the parity metrics?

Emulates, inside the CPU oracle, pointwise convolutions whose operands (activations, weights and
upstream gradients) carry `nb` significant bits (bf16 hi+lo pair: nb = 16; hi+mid+lo: nb = 24 = fp32)
and reports the parity report of tests/parity.py for each golden training case.

    python tests/exp_split_precision.py 16 train_M_2x4x32_s1 train_M_8x4x64_s2
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import x3d_oracle as xo  # noqa: E402
from x3dhip import synthetic  # noqa: E402
import parity  # noqa: E402

NB = int(sys.argv[1])
MODE = "all"                       # all: forward + both backward GEMMs; bwd: backward GEMMs only; wgrad: weight gradient only
args = sys.argv[2:]
if args and args[0] in ("all", "bwd", "wgrad", "dgrad"):
    MODE = args.pop(0)
cases = args or ["train_M_2x4x32_s1"]
_conv3d = F.conv3d


def rnd(t):
    if NB >= 24:
        return t
    drop = 24 - NB
    i = t.contiguous().view(torch.int32)
    i = (i + (1 << (drop - 1))) & ~((1 << drop) - 1)
    return i.view(torch.float32)


class PwSplit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride):
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        if MODE != "all":
            return _conv3d(x, w, stride=stride)
        return _conv3d(rnd(x), rnd(w), stride=stride)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gr = rnd(g)
        with torch.enable_grad():
            xe = x.detach().requires_grad_(True)
            we = w.detach().requires_grad_(True)
            dx_e, dw_e = torch.autograd.grad(_conv3d(xe, we, stride=ctx.stride), [xe, we], g)
            xr = rnd(x).detach().requires_grad_(True)
            wr = rnd(w).detach().requires_grad_(True)
            dx_r, dw_r = torch.autograd.grad(_conv3d(xr, wr, stride=ctx.stride), [xr, wr], gr)
        dx = dx_r if MODE in ("all", "bwd", "dgrad") else dx_e
        dw = dw_r if MODE in ("all", "bwd", "wgrad") else dw_e
        return dx, dw, None


def conv3d_patched(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if groups == 1 and tuple(w.shape[2:]) == (1, 1, 1) and b is None and x.dtype == torch.float32 and x.shape[-1] > 1:
        return PwSplit.apply(x, w, stride)
    return _conv3d(x, w, b, stride, padding, dilation, groups)


xo.F.conv3d = conv3d_patched
gold = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)
for case in cases:
    g = np.load(os.path.join(gold, case + ".npz"), allow_pickle=True)
    B, T, H, S = [int(v) for v in g["shape"]]
    sd = synthetic.procedural_state_dict(xo.state_template("M", 400, S), int(g["seed"][0]))
    x = synthetic.synthetic_clips(B, T, H, H, seed=int(g["seed"][1]))
    y = synthetic.synthetic_labels(B, seed=int(g["seed"][1]))
    logits, loss, grads, new_stats = xo.train_step_grads(x, y, sd, "M", S)
    e = parity.rel(logits[:, :, 0].numpy(), g["logits"])
    el = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
    try:
        rep = parity.check_grads({k: v.numpy() for k, v in grads.items()}, g, synthetic.gradient_sketch)
        status = "PASS"
    except AssertionError as ex:
        rep = ex.args[0] if ex.args else {}
        status = "FAIL"
    print(case, "nb", NB, "logits %.2e loss %.2e" % (e, el), status)
    if isinstance(rep, dict):
        print("   ", parity.fmt(rep))
    else:
        print("   ", rep)
