"""Debug aid (not a test): run the same training step twice on freshly built models with the caching allocator's free
blocks poisoned (NaN / garbage), compare logits / loss / gradients bitwise and report the first differing parameter."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch  # noqa: E402
import x3d  # noqa: E402
from oracle import x3d_oracle as xo  # noqa: E402
from x3dhip import synthetic  # noqa: E402

dev = torch.device("cuda:0")
poison = sys.argv[1] if len(sys.argv) > 1 else "nan"
S = 2
sd = synthetic.procedural_state_dict(xo.state_template("M", 400, S), 1)
x = synthetic.synthetic_clips(4, 4, 64, 64, seed=5).to(dev)
y = synthetic.synthetic_labels(4, seed=5).to(dev)


def poison_pool():
    blocks = [torch.empty(n, device=dev) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16) for _ in range(3)]
    for b in blocks:
        if poison == "nan":
            b.fill_(float("nan"))
        else:
            b.uniform_(-1e3, 1e3)
    del blocks


def run():
    net = x3d.generate_model("M", n_classes=400, dropout=0.0, base_bn_splits=S)
    net.load_state_dict(sd)
    net.to(dev).train(True)
    poison_pool()
    logits = net(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().clone(), loss.item(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


a = run()
b = run()
print("poison", poison, "logits equal", torch.equal(a[0], b[0]), "loss", a[1], b[1], "nan in logits", bool(torch.isnan(a[0]).any()))
bad = [k for k in a[2] if not torch.equal(a[2][k], b[2][k])]
nan = [k for k in a[2] if torch.isnan(a[2][k]).any()]
print("params differing:", len(bad), bad[:12])
print("params with NaN:", len(nan), nan[:12])

# ---- Trainer paths: single captured graph vs split graphs, two runs each
from x3dhip.trainer import Trainer  # noqa: E402


def run_trainer(split):
    net = x3d.generate_model("M", n_classes=400, dropout=0.0, base_bn_splits=S)
    net.load_state_dict(sd)
    net.to(dev).train(True)
    tr = Trainer(net, lr=0.05, use_graph=True, force_split=split)
    poison_pool()
    out = []
    for _ in range(2):
        loss, logits = tr.train_step(x, y)
        torch.cuda.synchronize()
        out.append((float(loss), tr.fp.grad.clone()))
    return out


runs = {"single-1": run_trainer(False), "single-2": run_trainer(False), "split-1": run_trainer(True), "split-2": run_trainer(True)}
ref = runs["single-1"]
for k, v in runs.items():
    print(k, "step1 loss %.9f grads==single-1: %s | step2 loss %.9f grads==: %s" % (
        v[0][0], torch.equal(v[0][1], ref[0][1]), v[1][0], torch.equal(v[1][1], ref[1][1])))
g0 = ref[0][1]
for k, v in runs.items():
    d = (v[0][1] - g0).abs()
    if d.max() > 0:
        idx = int(d.argmax())
        print("  ", k, "step-1 grad max abs diff %.3e at flat index %d (rel %.2e)" % (float(d.max()), idx, float(d.max() / g0.abs().max())))
