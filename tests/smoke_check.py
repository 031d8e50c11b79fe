"""Smoke check used by __graft_entry__.smoke(): one tiny X3D-M train step through the HIP
path on the GPU, verified against the CPU oracle on the same procedural weights and clip."""
import torch

from oracle import x3d_oracle as xo
from tests import parity
from x3dhip import synthetic


def run(dev, B=4, T=4, H=64, S=2):
    import x3d
    sd = synthetic.procedural_state_dict(xo.state_template("M", 400, S), 0)
    net = x3d.generate_model("M", n_classes=400, dropout=0.0, base_bn_splits=S)
    net.load_state_dict(sd)
    net = net.to(dev).train(True)
    x = synthetic.synthetic_clips(B, T, H, H)
    y = synthetic.synthetic_labels(B)
    logits = net(x.to(dev))
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    ref_logits, ref_loss, ref_grads, _ = xo.train_step_grads(x, y, sd, "M", S)
    e_log = parity.rel(logits.detach().cpu().numpy(), ref_logits.numpy())
    e_loss = abs(loss.item() - ref_loss.item()) / abs(ref_loss.item())
    got = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters())).item()
    ref = torch.sqrt(sum((g.double() ** 2).sum() for g in ref_grads.values())).item()
    e_g = abs(got - ref) / ref
    print("smoke: X3D-M (%d,3,%d,%d,%d) splits=%d  logits rel %.2e  loss rel %.2e  grad-norm rel %.2e"
          % (B, T, H, H, S, e_log, e_loss, e_g))
    assert e_log < 1e-3 and e_loss < 1e-3 and e_g < 1e-3
    return e_log, e_loss, e_g
