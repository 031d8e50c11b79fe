"""Per-workgroup timeline of pw6_kernel (stage 3-4 pointwise forward; needs the -DX3D_TRACE build libx3dhip_trace.so)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from x3dhip import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libx3dhip_trace.so")
from x3dhip import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = {"c1_3": (8, 96, 216, 16, 14), "c3_3": (8, 216, 96, 16, 14), "c1_4": (8, 192, 432, 16, 7), "c3_4": (8, 432, 192, 16, 7)}
which = sys.argv[1] if len(sys.argv) > 1 else "c1_3"
N, Ci, Co, T, H = shapes[which]
x = torch.randn(N, Ci, T, H, H, device=dev)
w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
pre = torch.rand(N, Ci, 2, device=dev)
wp = ops.pw_pack(w)
for _ in range(5):
    ops.pw_fwd(x, w, pre=pre, pre_act=2, wp=wp)
torch.cuda.synchronize()
buf = np.zeros(16384 * 8, dtype=np.uint64)
rc = _lib.lib().x3d_debug_p6trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
tr = buf.reshape(-1, 8)
tr = tr[tr[:, 0] > 0].astype(np.int64)
t0 = tr[:, 0].min()
rel = (tr - t0) * 10
print("case", which, "workgroups", len(tr), "span %.1f us" % (rel[:, 6].max() / 1000))
names = ["start", "loads issued", "loads arrived", "staged", "barrier", "mfma done", "stores done"]
for i in range(7):
    v = rel[:, i]
    v = v[v >= 0]
    print("  %-14s p10 %6d p50 %6d p90 %6d max %6d ns" % (names[i], np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
ok = tr[:, 5] > 0
d = lambda a, b: int(np.median(rel[ok, a] - rel[ok, b]))
print("  per workgroup (p50): issue %d, round trip %d, staging %d, barrier %d, mfma %d, stores %d, total %d ns"
      % (d(1, 0), d(2, 1), d(3, 2), d(4, 3), d(5, 4), d(6, 5), d(6, 0)))
