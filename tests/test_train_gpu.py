"""Training driver on the GPU: the multigrid loop (all long cycles of a compact schedule, BN-split
switching, LR rules), hipGraph replay vs eager launches, checkpoint round trip in the reference's
format, fused SGD vs torch.optim.SGD on the real model."""
import os

import pytest
import torch

from oracle import x3d_oracle as xo
from x3dhip import synthetic

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_multigrid_loop_runs_through_all_long_cycles(tmp_path, capsys):
    _dev()
    import train_x3d_kinetics_multigrid as tr
    save = str(tmp_path / "ck_")
    steps, cps = tr.run(init_lr=0.01, warmup_steps=5, max_epochs=4, batch_size=2, steps=0, max_steps_run=38,
                        iterations_per_epoch=10, save_model=save, save_every=20, use_graph=True, log_every=10,
                        val_every=15, val_batches=1, val_batch_size=1)
    out = capsys.readouterr().out
    assert steps == 38 and cps > 0
    assert out.count(" val after step") == 2            # validation phase (aggregate BN, 3-crop eval) interleaved, training resumes
    # banners for long cycles 0,1,2,3 of phase 1 and the restart of phase 2 ...
    assert out.count("*****") >= 2 * 5
    assert "BN_splits 8 long_ind 0" in out and "BN_splits 1 long_ind 3" in out and "long_ind -1" in out
    assert os.path.exists(save + "000020.pt")
    ck = torch.load(save + "000020.pt", map_location="cpu")
    assert set(ck.keys()) == {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "long_ind"}
    assert len(ck["model_state_dict"]) == 820
    assert len(ck["optimizer_state_dict"]["state"]) == 316
    # resume from the checkpoint (split-BN re-shaped before load, train...:166-173)
    steps2, _ = tr.run(init_lr=0.01, warmup_steps=5, max_epochs=4, batch_size=2, steps=20, max_steps_run=4,
                       iterations_per_epoch=10, load_ckpt=save + "000020.pt", save_every=0, use_graph=False)
    assert steps2 == 24


def _run_mode(mode, nsteps, sd, x, y, dev):
    import x3d
    from x3dhip.trainer import Trainer
    net = x3d.generate_model("M", dropout=0.0, base_bn_splits=1)
    net.load_state_dict(sd)
    net.to(dev).train(True)
    if mode == "torch":
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-5)
        for _ in range(nsteps):
            opt.zero_grad()
            loss = torch.nn.functional.cross_entropy(net(x), y)
            loss.backward()
            opt.step()
    else:
        tr = Trainer(net, lr=0.05, use_graph=(mode == "graph"))
        for _ in range(nsteps):
            loss, _ = tr.train_step(x, y)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).double().cpu()
    st = net.state_dict()
    return (float(loss.detach()), flat, st["layer2.1.bn2.split_bn.running_var"].cpu(),
            int(st["bn5.split_bn.num_batches_tracked"]))


def test_graph_replay_equals_eager_and_torch_sgd():
    dev = _dev()
    sd = synthetic.procedural_state_dict(xo.state_template("M", 400, 1), 0)
    x = synthetic.synthetic_clips(4, 4, 48, 48).to(dev)
    y = synthetic.synthetic_labels(4).to(dev)
    (l0, f0, r0, n0), (l1, f1, r1, n1) = [_run_mode(m, 3, sd, x, y, dev) for m in ("eager", "graph")]
    assert n0 == n1 == 3
    assert abs(l0 - l1) < 1e-5 * abs(l0) and torch.allclose(f0, f1, rtol=0, atol=1e-6)
    assert torch.allclose(r0, r1, rtol=1e-5)
    # fused SGD over the flat buffers == torch.optim.SGD, checked after TWO steps (the second uses
    # the momentum buffer); longer trajectories diverge through fp32 ReLU flips (tests/parity.py)
    (l0, f0, _, _), (l2, f2, _, _) = [_run_mode(m, 2, sd, x, y, dev) for m in ("eager", "torch")]
    assert abs(l0 - l2) < 1e-3 * abs(l0)
    assert ((f0 - f2).norm() / f2.norm()).item() < 1e-3     # update size is ~0.15 relative per step


def test_validate_matches_oracle_multi_crop():
    """validate(): aggregate split-BN stats, eval forward of b*n temporal crops, crop-averaged softmax / logits
    (train_x3d_kinetics_multigrid.py:203-206, 239-266) against the same protocol evaluated with the CPU oracle."""
    import torch.nn.functional as F
    import x3d as resnet_x3d
    import train_x3d_kinetics_multigrid as tr
    from oracle import x3d_oracle as xo
    from x3dhip import synthetic
    dev = torch.device("cuda:0")
    S = 2
    sd = synthetic.procedural_state_dict(xo.state_template("M", 400, S), 3)
    # make the split running statistics differ per split so that the aggregation matters
    for k in list(sd):
        if "split_bn.running_mean" in k:
            sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(len(k)))
        if "split_bn.running_var" in k:
            sd[k] = sd[k] * (1 + 0.2 * torch.rand(sd[k].shape, generator=torch.Generator().manual_seed(len(k) + 1)))
    model = resnet_x3d.generate_model(x3d_version="M", n_classes=400, dropout=0.5, base_bn_splits=S)
    model.load_state_dict(sd)
    model.to(dev)
    b, n, T, H = 2, 3, 4, 48
    batches = []
    for i in range(2):
        x = synthetic.synthetic_clips(b * n, T, H, H, seed=50 + i).view(b, n, 3, T, H, H)
        y = synthetic.synthetic_labels(b, seed=60 + i).view(b)
        batches.append((x, y))
    loss, acc, seen = tr.validate(model, [(x.to(dev), y.to(dev)) for x, y in batches])
    assert seen == 2 * b and not model.training
    # oracle: aggregate, eval forward, same reductions
    sd2 = dict(sd)
    sd2.update(xo.aggregate_sub_bn(sd, S))
    tot, corr = 0.0, 0
    for x, y in batches:
        lg = xo.forward(x.view(b * n, 3, T, H, H), sd2, "M", S, training=False)         # [b*n, 400, 1]
        lg = lg.view(b, n, 400, 1)
        sm = F.softmax(lg, dim=2).mean(1)
        lgm = lg.mean(1)
        tot += float(F.cross_entropy(lgm, y.view(b, 1)))
        corr += int((sm.max(1)[1] == y.view(b, 1)).sum())
    assert abs(loss - tot / 2) / (tot / 2) < 1e-3
    assert acc == corr / (2 * b)


def test_split_graph_backward_equals_single_graph(monkeypatch):
    """The two-graph backward used for data-parallel overlap (graph A: forward + head + layer4/3 backward, graph B:
    layer2/1 + stem backward, all-reduce bucket in between) gives the same loss and bit-identical gradients as the
    single captured graph."""
    import x3d as resnet_x3d
    from oracle import x3d_oracle as xo
    from x3dhip import synthetic
    from x3dhip.trainer import Trainer
    dev = torch.device("cuda:0")
    x = synthetic.synthetic_clips(4, 4, 64, 64, seed=5).to(dev)
    y = synthetic.synthetic_labels(4, seed=5).to(dev)
    res = []
    for split in (False, True):
        if split:
            monkeypatch.setenv("X3D_FORCE_SPLIT", "1")
        else:
            monkeypatch.delenv("X3D_FORCE_SPLIT", raising=False)
        model = resnet_x3d.generate_model(x3d_version="M", n_classes=400, dropout=0.0, base_bn_splits=2)
        model.load_state_dict(synthetic.procedural_state_dict(xo.state_template("M", 400, 2), 1))
        model.to(dev).train(True)
        tr = Trainer(model, lr=0.05, use_graph=True)
        assert tr._overlap() == split
        for _ in range(2):                       # capture + one replay
            loss, logits = tr.train_step(x, y)
        torch.cuda.synchronize()
        res.append((float(loss), tr.fp.grad.clone(), tr.fp.flat.clone(), int(model.state_dict()["bn1.split_bn.num_batches_tracked"])))
    # the loss VALUE comes out of torch's cross-entropy reduction, which is not bitwise reproducible run to run on this
    # stack (identical logits and gradients, last-bit differences in the scalar: tests/debug_determinism.py); everything
    # produced by the HIP path -- gradients, updated weights -- must be bit-identical
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0])
    assert torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2])
    assert res[0][3] == res[1][3] == 2


def test_training_reduces_loss_on_a_fixed_batch():
    """End-to-end sanity of the whole step (forward, backward, split-precision GEMMs, fused SGD, BN running stats,
    hipGraph replay): 40 steps on one fixed synthetic batch must drive the loss far below log(400) = 5.99."""
    import x3d as resnet_x3d
    from x3dhip import synthetic
    from x3dhip.trainer import Trainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = resnet_x3d.generate_model(x3d_version="M", n_classes=400, dropout=0.0, base_bn_splits=1).to(dev).train(True)
    tr = Trainer(model, lr=0.02, momentum=0.9, weight_decay=5e-5, use_graph=True)
    x = synthetic.synthetic_clips(8, 4, 64, 64, seed=3).to(dev)
    y = synthetic.synthetic_labels(8, seed=3).to(dev)
    losses = []
    for _ in range(40):
        loss, logits = tr.train_step(x, y)
        losses.append(float(loss))
    assert all(l == l for l in losses), "NaN in the loss"
    assert losses[0] > 4.5 and losses[-1] < 0.5 * losses[0], losses[::8]
    acc = float((logits.argmax(1) == y).float().mean())
    assert acc >= 0.75, (acc, losses[::8])
