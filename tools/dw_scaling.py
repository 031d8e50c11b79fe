"""Time the channelwise kernels of the small planes (stages 3-4) against the number of workgroups: does the chip have spare
issue slots at the base shape's 432 / 216 workgroups?  (Graph-timed, one kernel per replay.)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "x3d-multigrid_amd"))
import torch
from x3dhip import ops

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


for (C, T, H) in ((216, 16, 14), (432, 16, 7), (108, 16, 28)):
    for N in (8, 16, 32):
        x = torch.randn(N, C, T, H, H, device=dev)
        w = torch.randn(C, 1, 3, 3, 3, device=dev)
        pre = torch.randn(N, C, 2, device=dev)
        cb = torch.randn(N, C, 3, device=dev)
        tf = timed(lambda: ops.dw333_fwd(x, w, pre=pre, pre_act=1))
        tb = timed(lambda: ops.dw333_bwd(x, x, cb, w, x, pre=pre, pre_act=1, reduce=False))
        print("C=%d T=%d H=%d N=%d  fwd %.1f us  bwd %.1f us   (per sample fwd %.2f bwd %.2f)" % (C, T, H, N, tf, tb, tf / N, tb / N))
