"""Micro-benchmark of single kernels at the X3D-M layer shapes (for rocprofv3 --pmc runs)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
from x3dhip import ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "pw_l4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def t(fn):
    """us per launch: `reps` launches captured into one hipGraph and replayed (a Python/ctypes call costs
    ~25 us, more than most of these kernels run, so eager back-to-back timing measures the host)."""
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


cases = {
    "pw_l4": (8, 432, 192, 16, 7, 7),
    "pw_l4b": (8, 192, 432, 16, 7, 7),
    "pw_l3": (8, 216, 96, 16, 14, 14),
    "pw_l3b": (8, 96, 216, 16, 14, 14),
    "pw_l2": (8, 108, 48, 16, 28, 28),
    "pw_l2b": (8, 48, 108, 16, 28, 28),
    "pw_l1": (8, 24, 54, 16, 56, 56),
    "pw_l1b": (8, 54, 24, 16, 56, 56),
    "pw_l10": (8, 24, 54, 16, 112, 112),
}
if which.startswith("pw"):
    N, Ci, Co, T, H, W = cases[which]
    x = torch.randn(N, Ci, T, H, W, device=dev)
    w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
    pre = torch.rand(N, Ci, 2, device=dev)
    y = torch.empty(N, Co, T, H, W, device=dev)
    wp = ops.pw_pack(w)
    us = t(lambda: ops.pw_fwd(x, w, pre=pre, pre_act=2, out=y, wp=wp))
    fl = 2.0 * N * Ci * Co * T * H * W
    by = 4.0 * N * (Ci + Co) * T * H * W
    print("%s fwd(affine+swish): %.1f us  %.2f TFLOP/s  %.0f GB/s" % (which, us, fl / us / 1e6, by / us / 1e3))
    us = t(lambda: ops.pw_fwd(x, w, out=y, wp=wp))
    print("%s fwd(raw): %.1f us  %.2f TFLOP/s  %.0f GB/s" % (which, us, fl / us / 1e6, by / us / 1e3))
elif which.startswith("dg"):
    # backward-data of a conv with Cin -> Cout (GEMM K = Cout, M = Cin), swish backward in the epilogue
    N, Ci, Co, T, H, W = cases["pw" + which[2:]]
    x = torch.randn(N, Ci, T, H, W, device=dev)
    g = torch.randn(N, Co, T, H, W, device=dev)
    a = torch.randn(N, Co, T, H, W, device=dev)
    cb = torch.rand(N, Co, 3, device=dev)
    pre = torch.rand(N, Ci, 2, device=dev)
    w = torch.randn(Co, Ci, device=dev) / Co ** 0.5
    wpt = ops.pw_pack(w, transposed=True)
    out = torch.empty_like(x)
    us = t(lambda: ops.pw_bwd_data(g, a, cb, w, x=x, pre=pre, pre_act=2, out=out, wpt=wpt))
    fl = 2.0 * N * Ci * Co * T * H * W
    by = 4.0 * N * (2 * Ci + 2 * Co) * T * H * W
    print("%s dgrad(bn-bwd in, swish-bwd out): %.1f us  %.2f TFLOP/s  %.0f GB/s" % (which, us, fl / us / 1e6, by / us / 1e3))
elif which.startswith("wg"):
    N, Ci, Co, T, H, W = cases["pw" + which[2:]]
    x = torch.randn(N, Ci, T, H, W, device=dev)
    g = torch.randn(N, Co, T, H, W, device=dev)
    a = torch.randn(N, Co, T, H, W, device=dev)
    cb = torch.rand(N, Co, 3, device=dev)
    pre = torch.rand(N, Ci, 2, device=dev)
    us = t(lambda: ops.pw_bwd_weight(g, a, cb, x, (Co, Ci), pre=pre, pre_act=2))
    fl = 2.0 * N * Ci * Co * T * H * W
    by = 4.0 * N * (Ci + 2 * Co) * T * H * W
    print("%s wgrad(+reduce): %.1f us  %.2f TFLOP/s  %.0f GB/s" % (which, us, fl / us / 1e6, by / us / 1e3))
elif which.startswith("dw"):
    shapes = {"dw_l1": (8, 54, 16, 56, 56, 1), "dw_l10": (8, 54, 16, 112, 112, 2), "dw_l2": (8, 108, 16, 28, 28, 1),
              "dw_l3": (8, 216, 16, 14, 14, 1), "dw_l4": (8, 432, 16, 7, 7, 1)}
    N, C, T, H, W, s = shapes[which]
    x = torch.randn(N, C, T, H, W, device=dev)
    w = torch.randn(C, 1, 3, 3, 3, device=dev)
    pre = torch.rand(N, C, 2, device=dev)
    us = t(lambda: ops.dw333_fwd(x, w, stride=s, pre=pre))
    Ho = (H - 1) // s + 1
    by = 4.0 * N * C * T * (H * W + Ho * Ho)
    print("%s fwd: %.1f us  %.0f GB/s" % (which, us, by / us / 1e3))
    y, _ = ops.dw333_fwd(x, w, stride=s, pre=pre)
    g = torch.randn_like(y)
    cb = torch.rand(N, C, 3, device=dev)
    us = t(lambda: ops.dw333_bwd(g, y, cb, w, x, stride=s, pre=pre))
    print("%s bwd: %.1f us  %.0f GB/s (alg 2x(in+out))" % (which, us, 2 * by / us / 1e3))
