import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "x3d-multigrid_amd"))
import torch
from x3dhip import ops
dev = torch.device("cuda:0")
x = torch.randn(8, 3, 16, 224, 224, device=dev); w = torch.randn(24, 3, 1, 3, 3, device=dev)
for _ in range(3): ops.stem133_fwd(x, w)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): ops.stem133_fwd(x, w)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print("stem133_fwd %.1f us" % (e0.elapsed_time(e1) / 100 * 1000))
