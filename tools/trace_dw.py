"""Per-workgroup timeline of dw_fwd_kernel (needs the -DX3D_TRACE build: libx3dhip_trace.so); 100 MHz ticks."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from x3dhip import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libx3dhip_trace.so")
from x3dhip import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = {"l1": (8, 54, 16, 56, 56, 1), "l2": (8, 108, 16, 28, 28, 1), "l3": (8, 216, 16, 14, 14, 1), "l4": (8, 432, 16, 7, 7, 1)}
which = sys.argv[1] if len(sys.argv) > 1 else "l3"
N, C, T, H, W, s = shapes[which]
x = torch.randn(N, C, T, H, W, device=dev)
w = torch.randn(C, 1, 3, 3, 3, device=dev)
pre = torch.rand(N, C, 2, device=dev)
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
if bwd:
    y, _ = ops.dw333_fwd(x, w, stride=s, pre=pre)
    gy, cb = torch.randn_like(y), torch.rand(N, C, 3, device=dev)
    for _ in range(5):
        ops.dw333_bwd(gy, y, cb, w, x, stride=s, pre=pre, reduce=False)
else:
    for _ in range(5):
        ops.dw333_fwd(x, w, stride=s, pre=pre)
torch.cuda.synchronize()
buf = np.zeros(16384 * 8, dtype=np.uint64)
rc = _lib.lib().x3d_debug_dwtrace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
tr = buf.reshape(-1, 8)
tr = tr[tr[:, 0] > 0].astype(np.int64)
t0 = tr[:, 0].min()
rel = (tr - t0) * 10
print("case", which, "bwd" if bwd else "fwd", "workgroups", len(tr), "span %.1f us" % (rel[:, 7].max() / 1000))
names = ["start", "prologue done", "step5 begin", "step5 stencil done", "step5 LDS staged", "step5 barrier passed", "loop done", "end"]
for i in range(8):
    v = rel[:, i]
    print("  %-22s p10 %6d p50 %6d p90 %6d max %6d ns" % (names[i], np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
d = rel[:, 3] - rel[:, 2]; print("  step5: stencil %d ns (p50), staging wait %d, store+barrier %d, whole step %d; loop/16 = %d" % (
    np.median(d), np.median(rel[:, 4] - rel[:, 3]), np.median(rel[:, 5] - rel[:, 4]), np.median(rel[:, 5] - rel[:, 2]),
    np.median(rel[:, 6] - rel[:, 1]) / T))
