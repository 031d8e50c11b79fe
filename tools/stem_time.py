"""Graph-timed stem kernels at the base shape (8 x 3 x 16 x 224 x 224): python tools/stem_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
from x3dhip import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(8, 3, 16, 224, 224, device=dev)
w = torch.randn(24, 3, 1, 3, 3, device=dev)
dy = torch.randn(8, 24, 16, 112, 112, device=dev)


def t(fn, reps=20):
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


print("stem133_fwd        %.1f us" % t(lambda: ops.stem133_fwd(x, w)))
print("stem133_bwd_weight %.1f us (incl. group sum)" % t(lambda: ops.stem133_bwd_weight(x, dy, w.shape)))
