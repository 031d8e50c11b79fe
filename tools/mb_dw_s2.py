"""The four stride-2 channelwise launches of X3D-M at the base shape (first block of every stage), one kernel per graph replay:
time and effective HBM rate (algorithmic bytes: forward x + y, backward g + a + x + dx)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "x3d-multigrid_amd"))
import torch
from x3dhip import ops, _lib

dev = torch.device("cuda:0")
spec = sys.argv[1] if len(sys.argv) > 1 else ""
opts = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1000)
    return best


N = 8
with _lib.options(**opts):
    for (C, T, H) in ((54, 16, 112), (108, 16, 56), (216, 16, 28), (432, 16, 14)):
        x = torch.randn(N, C, T, H, H, device=dev)
        w = torch.randn(C, 1, 3, 3, 3, device=dev)
        pre = torch.randn(N, C, 2, device=dev)
        y, _ = ops.dw333_fwd(x, w, stride=2, pre=pre, pre_act=1)
        cb = torch.randn(N, C, 3, device=dev)
        gy = torch.randn_like(y)
        tf = timed(lambda: ops.dw333_fwd(x, w, stride=2, pre=pre, pre_act=1))
        tb = timed(lambda: ops.dw333_bwd(gy, y, cb, w, x, stride=2, pre=pre, pre_act=1, reduce=False))
        bf = 4 * (x.numel() + y.numel()); bb = 4 * (2 * x.numel() + 2 * y.numel())
        print("C=%3d H=%3d  fwd %6.1f us (%4.2f TB/s)  bwd %6.1f us (%4.2f TB/s)  %s" % (C, H, tf, bf / tf * 1e-6, tb, bb / tb * 1e-6, spec), flush=True)
