"""What does a dependent launch of a TINY kernel cost inside a replayed hipGraph?  (round 4)
torch elementwise kernels on 64 floats, the library's finalize kernels on realistic stage-3 statistics, same kernel repeated
vs different kernels alternating, producer data written by the previous kernel vs untouched."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
from x3dhip import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, reps):
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
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (10 * reps) * 1000


x = torch.zeros(64, device=dev)
y = torch.zeros(64, device=dev)
print("torch add_ x400 (same kernel, same 256 B)          %.2f us / node" % t(lambda: x.add_(1.0), 400))
def alt():
    x.add_(1.0); x.mul_(0.5)
print("torch add_ / mul_ alternating                      %.2f us / node" % (t(alt, 200) / 2))
def alt3():
    x.add_(1.0); torch.sin_(x); x.mul_(0.5); torch.tanh_(x)
print("torch add_/sin_/mul_/tanh_ (4 kernels)              %.2f us / node" % (t(alt3, 100) / 4))
big = torch.zeros(8, 216, 3136, device=dev)
def bigsmall():
    big.add_(1.0); x.add_(1.0)
tb = t(lambda: big.add_(1.0), 50)
print("torch add_ on 21.7 MB                              %.2f us / node" % tb)
print("torch [21.7 MB add_, tiny add_] pair               %.2f us / pair (tiny costs %.2f)" % (t(bigsmall, 50), t(bigsmall, 50) - tb))

N, C, tiles, S, P = 8, 216, 2, 1, 3136
part = torch.randn(N, C, tiles, 2, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
rm = torch.zeros(S, C, device=dev); rv = torch.ones(S, C, device=dev)
print("bn_fwd_finalize x100 (same kernel, same inputs)     %.2f us / node" % t(lambda: ops.bn_fwd_finalize(part, S, P, gamma, beta, rm, rv), 100))
coef, save, _ = ops.bn_fwd_finalize(part, S, P, gamma, beta, rm, rv)
def fin2():
    ops.bn_fwd_finalize(part, S, P, gamma, beta, rm, rv); ops.bn_bwd_finalize(part, S, P, gamma, save)
print("bn_fwd_finalize / bn_bwd_finalize alternating       %.2f us / node" % (t(fin2, 50) / 2))
part98 = torch.randn(N, 96, 98, 2, device=dev)
g96 = torch.ones(96, device=dev); save96 = torch.rand(2, S, 96, device=dev) + 0.5
print("bn_bwd_finalize (96 ch x 98 tiles) x100             %.2f us / node" % t(lambda: ops.bn_bwd_finalize(part98, S, P, g96, save96), 100))
def prodcons():
    part98.add_(0.001); ops.bn_bwd_finalize(part98, S, P, g96, save96)
tp = t(lambda: part98.add_(0.001), 100)
print("[torch add_ on the partials, bn_bwd_finalize]       %.2f us / pair (producer alone %.2f)" % (t(prodcons, 50), tp))
