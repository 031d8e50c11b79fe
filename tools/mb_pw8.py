"""Whole-K pointwise kernels of stages 3-4 at the headline shape: non-persistent pw6 / pw7 (option no_pw8) against the
persistent producer / consumer kernel pw8, `reps` dependent launches captured in one hipGraph (a dependent launch of
the replayed graph costs ~4.7 us by itself; the figure here includes it).   python tools/mb_pw8.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
from x3dhip import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def t(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1000


LAYERS = [("l3 conv1 96->216 @14^2", 8, 96, 216, 16, 14, 0), ("l3 conv3 216->96 @14^2", 8, 216, 96, 16, 14, 2),
          ("l4 conv1 192->432 @7^2", 8, 192, 432, 16, 7, 0), ("l4 conv3 432->192 @7^2", 8, 432, 192, 16, 7, 2)]
for name, N, Ci, Co, T, H, act in LAYERS:
    x = torch.randn(N, Ci, T, H, H, device=dev)
    w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
    pre = (torch.rand(N, Ci, 2, device=dev) + 0.5) if act else None
    wp, wpt = ops.pw_pack(w), ops.pw_pack(w, transposed=True)
    y = torch.empty(N, Co, T, H, H, device=dev)
    part = torch.empty(N, Co, _lib.lib().x3d_pw_fwd_tiles(N, Ci, Co, T * H * H, 1, 1), 2, device=dev)
    res = {}
    for tag, opts in (("pw6 8w", dict(no_pw8=1, pw_waves16=0)), ("pw6", dict(no_pw8=1)), ("pw6 w16=3", dict(no_pw8=1, pw_waves16=3)), ("pw8", dict(pw8_max_k=224)), ("pw8 g512", dict(pw8_grid=512, pw8_max_k=224)), ("pw8 g128", dict(pw8_grid=128, pw8_max_k=224))):
        with _lib.options(**opts):
            res[tag] = (t(lambda: ops.pw_fwd(x, w, pre=pre, pre_act=act, out=y, partial=part, wp=wp)), _lib.last_kernel())
    print("%-26s fwd  " % name + "  ".join("%s %.2f us (%s)" % (k, v[0], v[1]) for k, v in res.items()), flush=True)
    # data gradient of the same conv (K = Cout, M = Cin): conv3-type (activation backward) and conv1-type (residual backward)
    g = torch.randn(N, Co, T, H, H, device=dev); a = torch.randn(N, Co, T, H, H, device=dev); cb = torch.rand(N, Co, 3, device=dev)
    xo_ = torch.relu(torch.randn(N, Ci, T, H, H, device=dev)); ex = torch.randn(N, Ci, T, H, H, device=dev)
    pre2 = torch.rand(N, Ci, 2, device=dev) + 0.5
    res = {}
    for tag, opts in (("pw7 8w", dict(no_pw8=1, pw_waves16=0)), ("pw7", dict(no_pw8=1)), ("pw7 w16=3", dict(no_pw8=1, pw_waves16=3))):
        with _lib.options(**opts):
            ta = t(lambda: ops.pw_bwd_data(g, a, cb, w, x=ex, pre=pre2, pre_act=2, wpt=wpt))
            ka = _lib.last_kernel()
            tb = t(lambda: ops.pw_bwd_data_res(g, a, cb, w, xo_, ex, addend=xo_, wpt=wpt))
            res[tag] = (ta, tb, ka)
    print("%-26s bwd  " % name + "  ".join("%s act %.2f res %.2f us (%s)" % (k, v[0], v[1], v[2]) for k, v in res.items()), flush=True)
