"""Debug aid (not a test): one golden training case on the HIP path with every freshly allocated buffer NaN-filled
(x3dhip.ops.set_poison), run several times in one process, compared bitwise between runs and per parameter against
the fp64 CPU oracle on the same inputs.  A slot some kernel sums but nobody wrote shows up as a NaN in a named
parameter; a race shows up as a run-to-run difference; the per-parameter table localises a wrong layer.

    python tests/debug_poison.py [case] [runs] [nopoison]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import x3d  # noqa: E402
from oracle import x3d_oracle as xo  # noqa: E402
from x3dhip import ops, synthetic  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "train_M_2x4x158_s2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
poison = not (len(sys.argv) > 3 and sys.argv[3] == "nopoison")
g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
B, T, H, S = [int(v) for v in g["shape"]]
version = case.split("_")[1]
dev = torch.device("cuda:0")
sd = synthetic.procedural_state_dict(xo.state_template(version, 400, S), int(g["seed"][0]))
x = synthetic.synthetic_clips(B, T, H, H, seed=int(g["seed"][1]))
y = synthetic.synthetic_labels(B, seed=int(g["seed"][1]))
ops.set_poison(poison)


def dirty_pool():
    """Leave large finite garbage in the caching allocator's free blocks (what a long test session does)."""
    blocks = [torch.empty(n, device=dev).uniform_(-1e3, 1e3) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18) for _ in range(3)]
    del blocks


def run():
    net = x3d.generate_model(version, n_classes=400, dropout=0.0, base_bn_splits=S)
    net.load_state_dict(sd)
    net.to(dev).train(True)
    if not poison:
        dirty_pool()
    logits = net(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().cpu(), loss.item(), {k: p.grad.detach().cpu() for k, p in net.named_parameters()}


res = [run() for _ in range(runs)]
a = res[0]
print("case", case, "poison", poison, "loss", [r[1] for r in res], "NaN in logits", bool(torch.isnan(a[0]).any()))
for i, r in enumerate(res[1:], 1):
    bad = [k for k in a[2] if not torch.equal(a[2][k], r[2][k])]
    print("run %d vs run 0: logits equal %s, %d parameters differ bitwise %s" % (i, torch.equal(a[0], r[0]), len(bad), bad[:10]))
nan = [k for k in a[2] if torch.isnan(a[2][k]).any()]
print("parameters with NaN gradients (%d): %s" % (len(nan), nan[:40]))

# fp64 oracle on the same inputs
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
_, loss64, g64, _ = xo.train_step_grads(x.double(), y, sd64, version, S)
print("oracle fp64 loss %.9f" % float(loss64))
names = list(g64.keys())
gn = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g64.values())))
for i, r in enumerate(res):
    errs = []
    for k in names:
        d = float((r[2][k].double() - g64[k]).norm())
        errs.append((d / (float(g64[k].norm()) + 1e-6 * gn), k))
    gg = float(torch.sqrt(sum((r[2][k].double() ** 2).sum() for k in names)))
    print("run %d: global norm rel err %.3e; median per-parameter err %.3e" % (i, abs(gg - gn) / gn, float(np.median([e for e, _ in errs]))))
    if i == 0 or gg != gg or abs(gg - gn) / gn > 1e-3:
        print("  per-parameter relative error, network order (every 1st / worst shown):")
        for e, k in errs:
            if e != e or e > 2e-2:
                print("   %-40s %.3e" % (k, e))
        print("  bn1.bias.grad[:7]", r[2]["bn1.bias"][:7].tolist())
        print("  oracle            ", g64["bn1.bias"][:7].tolist())
