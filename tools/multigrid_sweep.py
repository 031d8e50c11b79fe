"""BASELINE config 3: X3D-M on one GPU through the multigrid (B, T, H, W) shape table with per-GPU base batch 8
(long cycle x short cycle, cycle_batch_sampler.py + kinetics_multigrid.py:205-237): every distinct step shape of a
compressed schedule, timed with hipGraph replay, plus the long-cycle BN-split switches the training loop performs.

    python tools/multigrid_sweep.py [steps_per_shape]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
import cycle_batch_sampler as cbs  # noqa: E402
import train_x3d_kinetics_multigrid as tr  # noqa: E402
import x3d as resnet_x3d  # noqa: E402
from kinetics_multigrid import device_batch  # noqa: E402
from x3dhip.trainer import Trainer  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
base = 8
shapes, _, _ = tr.setup_data(base, 1, 1, 60, 0, 224, [256., 320.], 80, 5)
DT = torch.bfloat16 if "--bf16" in sys.argv else torch.float32      # --bf16: mixed-storage mode
model = resnet_x3d.generate_model(x3d_version='M', n_classes=400, dropout=0.5, base_bn_splits=max(1, base // tr.CONST_BN_SIZE),
                                  act_dtype=DT)
model.to(dev).train(True)
opt = Trainer(model, lr=0.0125, use_graph=True)
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
seen, rows, last_long = set(), [], -2
for step_no, (n_global, long_ind, (T, H)) in enumerate(shapes):
    if step_no >= 59:                        # the compressed schedule has 60 steps
        break
    if long_ind != last_long:
        splits = model.update_bn_splits_long_cycle(tr.LONG_CYCLE[long_ind])
        last_long = long_ind
    key = (n_global, T, H, long_ind)
    if key in seen:
        continue
    seen.add(key)
    x, y = device_batch(n_global, T, H, 400, dev, gen)
    for _ in range(3):                      # capture + warm replays
        opt.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(K):
        loss, _ = opt.train_step(x, y)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / K * 1e3
    rows.append((long_ind, n_global, T, H, splits, ms, n_global / ms * 1e3, float(loss)))
    print("long %d  B %3d T %2d H %3d  bn_splits %d  %7.2f ms/step  %7.1f clips/s  loss %.3f" % rows[-1], flush=True)
vox = [r[1] * r[2] * r[3] * r[3] for r in rows]
print("shapes %d; B*T*H*W min %.2fM max %.2fM (base 8*16*224^2 = %.2fM)" % (len(rows), min(vox) / 1e6, max(vox) / 1e6, 8 * 16 * 224 * 224 / 1e6))
