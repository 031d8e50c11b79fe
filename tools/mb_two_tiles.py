import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/x3d-multigrid_amd')
import torch
from x3dhip import ops, _lib
dev = torch.device('cuda:0')
def bench(fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for (N, Ci, Co, T, H) in [(8, 432, 192, 16, 7), (8, 216, 96, 16, 14), (8, 192, 432, 16, 7)]:
    x = torch.randn(N, Ci, T, H, H, device=dev); w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
    pre = torch.rand(N, Ci, 2, device=dev) + 0.5
    wp = ops.pw_pack(w); wpt = ops.pw_pack(w, transposed=True)
    g = torch.randn(N, Co, T, H, H, device=dev); a = torch.randn(N, Co, T, H, H, device=dev); cb = torch.rand(N, Co, 3, device=dev)
    xo_ = torch.relu(torch.randn(N, Ci, T, H, H, device=dev)); ex = torch.randn(N, Ci, T, H, H, device=dev)
    for k in (4096, 320):
        with _lib.options(pw_two_tiles_k=k):
            tf = bench(lambda: ops.pw_fwd(x, w, pre=pre, pre_act=2, wp=wp))
            tb = bench(lambda: ops.pw_bwd_data_res(g, a, cb, w, xo_, ex, wpt=wpt))
        print("Cin %3d Cout %3d P %5d two_tiles_k %4d: fwd %6.2f us (%s)  dgrad %6.2f us" % (Ci, Co, T*H*H, k, tf, _lib.last_kernel(), tb))
