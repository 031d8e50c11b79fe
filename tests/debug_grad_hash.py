"""sha256 over logits + all gradients of golden training cases (one line per case): run on different boxes / in different
processes to see whether the step is bitwise the same everywhere.   python tests/debug_grad_hash.py [case ...]"""
import hashlib
import os
import socket
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from x3dhip import synthetic  # noqa: E402
import test_model_gpu as tm  # noqa: E402

dev = torch.device("cuda:0")
cases = sys.argv[1:] or ["train_M_2x4x32_s1", "train_M_2x4x158_s2", "train_M_8x4x64_s2", "train_M_2x4x111_s1"]
print("host", socket.gethostname(), "device", torch.cuda.get_device_name(0), "cus", torch.cuda.get_device_properties(0).multi_processor_count)
for case in cases:
    g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    B, T, H, S = [int(v) for v in g["shape"]]
    net = tm._build(case.split("_")[1], S, dev, int(g["seed"][0]))
    net.train(True)
    x = synthetic.synthetic_clips(B, T, H, H, seed=int(g["seed"][1])).to(dev)
    y = synthetic.synthetic_labels(B, seed=int(g["seed"][1])).to(dev)
    hx = hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:12]
    hw = hashlib.sha256(b"".join(p.detach().cpu().numpy().tobytes() for p in net.parameters())).hexdigest()[:12]
    logits = net(x)
    loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    h.update(logits.detach().cpu().numpy().tobytes())
    for _, p in net.named_parameters():
        h.update(p.grad.detach().cpu().numpy().tobytes())
    tot = float(np.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in net.parameters())))
    print("%-22s clip %s weights %s  logits+grads %s  global norm err %.3e" % (
        case, hx, hw, h.hexdigest()[:16], abs(tot - float(g["grad_global_norm64"])) / float(g["grad_global_norm64"])), flush=True)
