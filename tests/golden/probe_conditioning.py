"""How well-posed is a golden training fixture?  (CPU, fp64 oracle -- test infrastructure only.)

The gradient of a ReLU network is a discontinuous function of its input: a pre-activation that crosses zero switches a unit,
and with batch statistics over few voxels (small clips: stage 4 of X3D-M sees 1 x 1 planes at 32^2) one switch moves whole
tensors by percents.  An fp32 implementation perturbs every pre-activation by ~1e-7 relative, so a fixture whose fp64
gradient JUMPS under perturbations of that size cannot be reproduced to 1e-3 by ANY fp32 evaluation order except by luck.
This script measures that: K random relative perturbations of the clip (default 3e-7, the size of fp32 rounding noise after
a few layers), fp64 forward + backward of the oracle, change of the global gradient norm and of the worst tensor.

    python tests/golden/probe_conditioning.py train_M_2x4x32_s1 [K] [eps]

Output: one line per perturbation and a summary line  "CONDITIONING <case> max_global <v> median_global <v>"; the table of
all training fixtures is tests/golden/conditioning.json (written with --json)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import x3d_oracle as xo  # noqa: E402
from x3dhip import synthetic  # noqa: E402


def probe(case, K=6, eps=3e-7):
    g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"), allow_pickle=True)
    B, T, H, S = [int(v) for v in g["shape"]]
    version = case.split("_")[1]
    sd0 = {k: (v.double() if v.is_floating_point() else v)
           for k, v in synthetic.procedural_state_dict(xo.state_template(version, 400, S), int(g["seed"][0])).items()}
    x = synthetic.synthetic_clips(B, T, H, H, seed=int(g["seed"][1])).double()
    y = synthetic.synthetic_labels(B, seed=int(g["seed"][1]))

    def grads(xin):
        sd = {k: (v.clone().requires_grad_(True) if xo.is_parameter(k) else v) for k, v in sd0.items()}
        loss = xo.loss_fn(xo.forward(xin, sd, version, S, True), y)
        loss.backward()
        return {k: v.grad.detach() for k, v in sd.items() if xo.is_parameter(k) and v.grad is not None}

    g0 = grads(x)
    tot0 = float(torch.sqrt(sum((v ** 2).sum() for v in g0.values())))
    ref = abs(tot0 - float(g["grad_global_norm64"])) / float(g["grad_global_norm64"])
    print("%s: oracle fp64 global norm vs fixture %.2e" % (case, ref), flush=True)
    out, outt = [], []
    per = {n: [] for n in g0}                      # per-tensor responses (round 4: tests/parity.py allows per NAME)
    for k in range(K):
        torch.manual_seed(1000 + k)
        gi = grads(x * (1 + eps * torch.randn_like(x)))
        tot = float(torch.sqrt(sum((v ** 2).sum() for v in gi.values())))
        for n in g0:
            per[n].append(float((gi[n] - g0[n]).norm() / (g0[n].norm() + 1e-300)))
        w = max((per[n][-1], n) for n in g0)
        out.append(abs(tot - tot0) / tot0)
        outt.append(w[0])
        print("  perturbation %d (%.0e): global norm %.3e   worst tensor %s %.3e" % (k, eps, out[-1], w[1], w[0]), flush=True)
    print("CONDITIONING %s max_global %.3e median_global %.3e" % (case, max(out), float(np.median(out))), flush=True)
    return {"case": case, "eps": eps, "K": K, "global": out, "max_global": max(out), "median_global": float(np.median(out)),
            "worst_tensor": outt, "max_tensor": max(outt),
            # median response of every tensor that moves by more than 1e-4 (the others: 0); order-free, keyed by name
            "tensor_median": {n: float(np.median(v)) for n, v in per.items() if float(np.median(v)) > 1e-4}}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = probe(args[0], int(args[1]) if len(args) > 1 else 6, float(args[2]) if len(args) > 2 else 3e-7)
    if "--json" in sys.argv:
        p = os.path.join(ROOT, "tests", "golden", "conditioning.json")
        tab = json.load(open(p)) if os.path.exists(p) else {}
        tab[res["case"]] = res
        json.dump(tab, open(p, "w"), indent=1, sort_keys=True)
