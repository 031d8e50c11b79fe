"""Robustness sweep: one training step (hipGraph capture + replay) of every model version at its native clip shape and a
few odd shapes; reports ms/step.  Catches launch-configuration errors (LDS limits, grid limits, odd sizes)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
import x3d  # noqa: E402
from x3dhip.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
cases = [("S", 8, 13, 160, 1), ("M", 8, 16, 224, 1), ("XL", 2, 16, 312, 1), ("L", 2, 16, 312, 1), ("M", 6, 5, 79, 2),
         ("M", 3, 16, 224, 1), ("XL", 4, 4, 111, 2), ("M", 16, 16, 158, 2)]
DT = torch.bfloat16 if "--bf16" in sys.argv else torch.float32      # --bf16: mixed-storage mode
for ver, B, T, H, S in cases:
    net = x3d.generate_model(ver, n_classes=400, dropout=0.5, base_bn_splits=S, act_dtype=DT).to(dev).train(True)
    tr = Trainer(net, lr=0.01, use_graph=True)
    x = torch.randn(B, 3, T, H, H, device=dev)
    y = torch.randint(0, 400, (B, 1), device=dev)
    for _ in range(2):
        loss, _ = tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        loss, _ = tr.train_step(x, y)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 5 * 1e3
    assert loss == loss
    print("X3D-%-2s B %2d T %2d H %3d splits %d %s: %7.2f ms/step  %7.1f clips/s  loss %.3f"
          % (ver, B, T, H, S, "bf16" if DT == torch.bfloat16 else "fp32", ms, B / ms * 1e3, float(loss)), flush=True)
    del tr, net
    torch.cuda.empty_cache()
