"""Prologue phases of dw_bwd_kernel (temporary stamps; needs the X3D_TRACE build with the prologue stamps of this experiment)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import numpy as np, torch
from x3dhip import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libx3dhip_trace.so")
from x3dhip import ops
dev = torch.device("cuda:0")
shapes = {"l1": (8, 54, 16, 56), "l2": (8, 108, 16, 28), "l3": (8, 216, 16, 14), "l4": (8, 432, 16, 7)}
for which in sys.argv[1:] or ["l3"]:
    N, C, T, H = shapes[which]
    x = torch.randn(N, C, T, H, H, device=dev); w = torch.randn(C, 1, 3, 3, 3, device=dev)
    pre = torch.rand(N, C, 2, device=dev); cb = torch.rand(N, C, 3, device=dev)
    y, _ = ops.dw333_fwd(x, w, pre=pre); gy = torch.randn_like(y)
    for form in ("cb", "stats"):
        stiles = 98 if H <= 14 else 65
        sp = torch.randn(N, C, stiles, 2, device=dev); gamma = torch.rand(C, device=dev) + 0.5
        save = torch.cat([torch.randn(1, 1, C, device=dev), torch.rand(1, 1, C, device=dev) + 0.5]).contiguous()
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        for _ in range(4):
            if form == "cb": ops.dw333_bwd(gy, y, cb, w, x, pre=pre, reduce=False)
            else: ops.dw333_bwd(gy, y, None, w, x, pre=pre, reduce=False, bn=(sp, T * H * H, gamma, save, dg, db))
        torch.cuda.synchronize()
        buf = np.zeros(16384 * 8, dtype=np.uint64)
        assert _lib.lib().x3d_debug_dwtrace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
        tr = buf.reshape(-1, 8); tr = tr[tr[:, 0] > 0].astype(np.int64)
        rel = (tr - tr[:, :1]) * 10          # ns since the workgroup's own start
        m = lambda i: int(np.median(rel[:, i]))
        print("%s %-5s wgs %4d | zero-fill+weights %4d | make_chunks %4d | finalize/barrier %4d | 2nd plane + staging + barrier %4d | first window %4d | prologue %4d | loop %5d | epilogue %4d ns" % (
            which, form, len(tr), m(2), m(3) - m(2), m(4) - m(3), m(5) - m(4), m(1) - m(5), m(1), m(6) - m(1), m(7) - m(6)), flush=True)
