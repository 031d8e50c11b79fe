"""Per-workgroup phase timeline of pw2_kernel (needs the -DX3D_TRACE build: libx3dhip_trace.so).
Timestamps are s_memrealtime ticks (100 MHz)."""
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
cases = {"l3b": (8, 96, 216, 16, 14, 14), "l3": (8, 216, 96, 16, 14, 14), "l4": (8, 432, 192, 16, 7, 7),
         "l4b": (8, 192, 432, 16, 7, 7)}
which = sys.argv[1] if len(sys.argv) > 1 else "l3b"
N, Ci, Co, T, H, W = cases[which]
x = torch.randn(N, Ci, T, H, W, device=dev)
w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
pre = torch.rand(N, Ci, 2, device=dev)
y = torch.empty(N, Co, T, H, W, device=dev)
wp = ops.pw_pack(w)
for _ in range(5):
    ops.pw_fwd(x, w, pre=pre, pre_act=2, out=y, wp=wp)
torch.cuda.synchronize()
buf = np.zeros(16384 * 8, dtype=np.uint64)
rc = _lib.lib().x3d_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
tr = buf.reshape(-1, 8)
tr = tr[tr[:, 0] > 0]
t0 = tr[:, 0].min()
rel = (tr[:, :7].astype(np.int64) - np.int64(t0)) * 10          # ns
print("case", which, "workgroups traced", len(tr), "span %.1f us" % (rel[:, 6].max() / 1000))
if os.environ.get("X3D_PW_NO_PERSIST") is None:
    # persistent kernel: [start, (item compute done, item epilogue done) x up to 3]
    print("case", which, "(pw4) workgroups", len(tr), "span %.1f us" % ((tr[:, 1:7].max() - t0) * 10 / 1000))
    for i in range(1, 7):
        col = tr[:, i].astype(np.int64)
        ok = col > 0
        if ok.any():
            v = (col[ok] - np.int64(t0)) * 10
            print("  stamp %d (%s item %d): n %4d  p10 %6d  p50 %6d  p90 %6d  max %6d ns" % (
                i, "compute done" if i % 2 else "epilogue done", (i - 1) // 2, ok.sum(), np.percentile(v, 10),
                np.percentile(v, 50), np.percentile(v, 90), v.max()))
    st = (tr[:, 0].astype(np.int64) - np.int64(t0)) * 10
    print("  start p50 %d p90 %d max %d" % (np.percentile(st, 50), np.percentile(st, 90), st.max()))
    sys.exit(0)
names = ["start", "fetch0 issued", "Cl barrier", "store0+barrier", "main loop", "epi loads issued", "end"]
d = np.diff(rel, axis=1)
for i in range(6):
    print("  phase %-18s mean %7.0f ns  p50 %7.0f  p90 %7.0f  max %7.0f" % (names[i + 1], d[:, i].mean(), np.percentile(d[:, i], 50),
                                                                         np.percentile(d[:, i], 90), d[:, i].max()))
dur = rel[:, 6] - rel[:, 0]
print("  WG duration mean %.0f ns p50 %.0f p90 %.0f max %.0f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), dur.max()))
st = np.sort(rel[:, 0])
print("  WG start times (ns) percentiles:", [int(np.percentile(st, p)) for p in (0, 10, 25, 50, 75, 90, 100)])
en = np.sort(rel[:, 6])
print("  WG end times (ns) percentiles:  ", [int(np.percentile(en, p)) for p in (0, 10, 25, 50, 75, 90, 100)])
xcc = (tr[:, 7] >> np.uint64(32)).astype(np.int64) & 0xF
hw = (tr[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("  distinct (xcc,se,cu):", len(u), " WGs per CU min/mean/max:", cnt.min(), cnt.mean(), cnt.max())
print("  WGs per XCC:", np.bincount(xcc, minlength=8))
