"""A/B of the channelwise stride-1 kernels under library options: base-shape planes of the four stages, one kernel per graph
replay.  usage: python tools/mb_dw.py [name=value[,name=value]]      e.g.  python tools/mb_dw.py dw_th=8"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "x3d-multigrid_amd"))
import torch
from x3dhip import ops, _lib

dev = torch.device("cuda:0")
spec = sys.argv[1] if len(sys.argv) > 1 else ""
opts = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}


def timed(fn, reps=20):
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
for (C, T, H) in ((54, 16, 56), (108, 16, 28), (216, 16, 14), (432, 16, 7)):
    x = torch.randn(N, C, T, H, H, device=dev)
    w = torch.randn(C, 1, 3, 3, 3, device=dev)
    pre = torch.randn(N, C, 2, device=dev)
    cb = torch.randn(N, C, 3, device=dev)
    f0 = timed(lambda: ops.dw333_fwd(x, w, pre=pre, pre_act=1))
    b0 = timed(lambda: ops.dw333_bwd(x, x, cb, w, x, pre=pre, pre_act=1, reduce=False))
    line = "C=%3d H=%3d  fwd %6.1f us  bwd %6.1f us" % (C, H, f0, b0)
    if opts:
        with _lib.options(**opts):
            f1 = timed(lambda: ops.dw333_fwd(x, w, pre=pre, pre_act=1))
            b1 = timed(lambda: ops.dw333_bwd(x, x, cb, w, x, pre=pre, pre_act=1, reduce=False))
        line += "   | %s: fwd %6.1f (%+.1f%%)  bwd %6.1f (%+.1f%%)" % (spec, f1, 100 * (f0 / f1 - 1), b1, 100 * (b0 / b1 - 1))
    print(line, flush=True)
