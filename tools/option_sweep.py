"""A/B of library options on the headline step (X3D-M, B=8, T=16, 224^2, hipGraph replay), one process, same tensors:
every setting re-captures the graph and times K replays + SGD.

    python tools/option_sweep.py [K] name=value[,name=value...] ...      e.g.  pw_nt4_min=512 fb_grid=256 wg_cpw=4,wg_cap=512
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
import x3d  # noqa: E402
from x3dhip import _lib, synthetic  # noqa: E402
from x3dhip.trainer import Trainer  # noqa: E402

args = sys.argv[1:]
K = int(args.pop(0)) if args and args[0].isdigit() else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = x3d.generate_model("M", n_classes=400, dropout=0.5, base_bn_splits=1).to(dev).train(True)
tr = Trainer(net, lr=0.05, use_graph=True)
x = synthetic.synthetic_clips(8, 16, 224, 224, seed=1234).to(dev)
y = synthetic.synthetic_labels(8, seed=1234).to(dev)


def timed():
    tr.invalidate_graphs()
    for _ in range(3):
        tr.step(x, y)
    xs, ys = tr.static_inputs(x.shape)
    xs.copy_(x)
    ys.copy_(y)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            tr.step(xs, ys)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    return 1e3 * best


base = timed()
print("%-44s %7.3f ms/step  %7.1f clips/s" % ("defaults", base, 8e3 / base), flush=True)
from x3dhip import engine  # noqa: E402
for spec in args:
    kv = dict(kv.split("=") for kv in spec.split(","))
    opts = {k: int(v) for k, v in kv.items() if not k.startswith("cfg.")}
    cfgs = {k[4:]: bool(int(v)) for k, v in kv.items() if k.startswith("cfg.")}       # schedule switches: cfg.wgrad_overlap=1
    old = {k: getattr(engine.cfg, k) for k in cfgs}
    for k, v in cfgs.items():
        setattr(engine.cfg, k, v)
    with _lib.options(**opts):
        t = timed()
    for k, v in old.items():
        setattr(engine.cfg, k, v)
    print("%-44s %7.3f ms/step  %7.1f clips/s  (%+.2f %%)" % (spec, t, 8e3 / t, 100 * (base / t - 1)), flush=True)
t = timed()
print("%-44s %7.3f ms/step  %7.1f clips/s  (%+.2f %%)" % ("defaults again", t, 8e3 / t, 100 * (base / t - 1)), flush=True)
