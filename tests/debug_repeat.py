"""Debug aid (not a test): the same training step many times in ONE process, every run compared bitwise with the first
(logits + all 316 gradients, on the device).  The product claims bitwise determinism (no atomics, fixed-order sums): any
run that differs is a race or a read of memory nobody wrote; the names of the differing parameters localise it.

    python tests/debug_repeat.py [case] [iterations] [--after-suite] [--perturb]

--after-suite: first run the mixed-storage + trainer GPU tests in this process (allocator / graph-pool / scratch state of a
               long pytest session);   --perturb: random junk allocations and a busy side stream between iterations.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
case = args[0] if args else "train_M_2x4x158_s2"
iters = int(args[1]) if len(args) > 1 else 200

if "--after-suite" in flags:
    import pytest
    rc = pytest.main(["-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                      os.path.join(ROOT, "tests", "test_mixed_storage_gpu.py"),
                      os.path.join(ROOT, "tests", "test_model_gpu.py") + "::test_train_step_vs_reference_golden"])
    print("in-process suite rc", rc, flush=True)

import x3d  # noqa: E402
from oracle import x3d_oracle as xo  # noqa: E402
from x3dhip import synthetic  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
B, T, H, S = [int(v) for v in g["shape"]]
version = case.split("_")[1]
dev = torch.device("cuda:0")
sd = synthetic.procedural_state_dict(xo.state_template(version, 400, S), int(g["seed"][0]))
x = synthetic.synthetic_clips(B, T, H, H, seed=int(g["seed"][1])).to(dev)
y = synthetic.synthetic_labels(B, seed=int(g["seed"][1])).to(dev)
net = x3d.generate_model(version, n_classes=400, dropout=0.0, base_bn_splits=S)
net.load_state_dict(sd)
net.to(dev).train(True)
names = [k for k, _ in net.named_parameters()]
rng = np.random.default_rng(0)
side = torch.cuda.Stream()
junk_a = torch.randn(2048, 2048, device=dev)


def step():
    for p in net.parameters():
        p.grad = None
    logits = net(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    return [logits.detach().clone()] + [p.grad.detach().clone() for p in net.parameters()]


ref = step()
torch.cuda.synchronize()
bad_runs = 0
for it in range(1, iters):
    if "--perturb" in flags:
        junk = [torch.empty(int(rng.integers(1, 1 << 22)), device=dev).fill_(float(rng.normal())) for _ in range(int(rng.integers(0, 6)))]
        with torch.cuda.stream(side):
            for _ in range(int(rng.integers(0, 4))):
                junk_a @ junk_a
        del junk
    cur = step()
    diff = torch.stack([(a != b).any() for a, b in zip(ref, cur)])
    if bool(diff.any()):
        bad_runs += 1
        idx = [i for i, d in enumerate(diff.tolist()) if d]
        what = ["logits" if i == 0 else names[i - 1] for i in idx]
        rel = []
        for i in idx[:6] + idx[-3:]:
            rel.append("%s:%.2e" % ("logits" if i == 0 else names[i - 1],
                                    float((ref[i].double() - cur[i].double()).norm() / ref[i].double().norm().clamp_min(1e-30))))
        print("iteration %d differs from iteration 0 in %d tensors; first (forward order) %s ... last %s | %s"
              % (it, len(idx), what[:4], what[-4:], " ".join(rel)), flush=True)
print("case %s: %d iterations, %d differ from the first" % (case, iters, bad_runs))
