"""Per-parameter gradient-norm error of one golden training case, in network order (where does a wrong backward start?).
usage: python tests/debug_golden_layers.py train_M_2x4x32_s1 [name=value ...library options]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import x3d  # noqa: E402
from x3dhip import _lib, synthetic  # noqa: E402

case = sys.argv[1]
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"), allow_pickle=True)
B, T, H, S = [int(v) for v in g["shape"]]
dev = torch.device("cuda:0")
torch.manual_seed(int(g["seed"][0]))
import test_model_gpu as tm  # noqa: E402
net = tm._build(case.split("_")[1], S, dev, int(g["seed"][0]))
net.train(True)
x = synthetic.synthetic_clips(B, T, H, H, seed=int(g["seed"][1])).to(dev)
y = synthetic.synthetic_labels(B, seed=int(g["seed"][1])).to(dev)
logits = net(x)
loss = torch.nn.CrossEntropyLoss()(logits, y)
loss.backward()
torch.cuda.synchronize()
names = list(g["grad_names"])
_tot = np.sqrt(sum(float(np.linalg.norm(p.grad.detach().cpu().numpy().astype(np.float64))) ** 2 for _, p in net.named_parameters()))
print("GLOBAL norm err %.3e (reference fp32 %.3e)  options %s" % (abs(_tot - float(g["grad_global_norm64"])) / float(g["grad_global_norm64"]),
      abs(float(g["grad_global_norm"]) - float(g["grad_global_norm64"])) / float(g["grad_global_norm64"]), sys.argv[2:]))
n64, n32 = g["grad_norms64"], g["grad_norms"]
gn = float(g["grad_global_norm64"])
grads = dict(net.named_parameters())
for i, k in enumerate(names):
    got = float(np.linalg.norm(grads[k].grad.detach().cpu().numpy().astype(np.float64)))
    e = abs(got - n64[i]) / (n64[i] + 1e-6 * gn)
    f = abs(n32[i] - n64[i]) / (n64[i] + 1e-6 * gn)
    flag = "  <<<" if e > 1e-3 + 3 * f else ""
    print("%-40s norm64 %.4e  err %.2e  (reference fp32 %.2e)%s" % (k, n64[i], e, f, flag))
