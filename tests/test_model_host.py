"""Host-side logic of the drop-in x3d module (no GPU): module tree / state_dict layout against
the reference's (golden keys.npz), BN-split switching, stat aggregation, loud failure off-GPU."""
import os

import numpy as np
import pytest
import torch

import x3d
from oracle import x3d_oracle as xo
from x3dhip import synthetic, _lib


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("v,s", [("M", 1), ("M", 4), ("XL", 2)])
def test_state_dict_layout_matches_reference(golden_dir, v, s):
    g = _golden(golden_dir, "keys")
    net = x3d.generate_model(v, n_classes=400, base_bn_splits=s)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["keys_%s_%d" % (v, s)])
    assert ["x".join(map(str, t.shape)) for t in sd.values()] == list(g["shapes_%s_%d" % (v, s)])
    assert [k for k, _ in net.named_parameters()] == list(g["params_%s_%d" % (v, s)])


def test_parameter_count_and_versions():
    assert sum(p.numel() for p in x3d.generate_model("M").parameters()) == 3794322
    assert sum(p.numel() for p in x3d.generate_model("S").parameters()) == 3794322
    assert sum(p.numel() for p in x3d.generate_model("XL").parameters()) == 11095904
    assert sum(p.numel() for p in x3d.generate_model("L").parameters()) == 6153432
    assert x3d.get_blocks("XL") == [5, 10, 25, 15]
    assert x3d.get_inplanes("M")[0] == (54, 24)
    assert x3d.Bottleneck.round_width(54) == 8 and x3d.Bottleneck.round_width(432) == 32


def test_reference_format_checkpoint_loads():
    net = x3d.generate_model("M", base_bn_splits=2)
    sd = synthetic.procedural_state_dict(xo.state_template("M", 400, 2), 0)
    missing, unexpected = net.load_state_dict(sd)
    assert not missing and not unexpected
    assert torch.equal(net.state_dict()["layer3.4.fc2.weight"], sd["layer3.4.fc2.weight"])


def test_update_bn_splits_and_aggregate(golden_dir):
    g = _golden(golden_dir, "keys")
    net = x3d.generate_model("M", base_bn_splits=2)
    r = net.update_bn_splits_long_cycle(4)
    assert r == int(g["update_ret"]) == 8
    assert list(net.state_dict()["bn1.split_bn.running_mean"].shape) == list(g["update_shape"])
    # fresh running stats after the switch (x3d.py:302)
    assert float(net.bn1.split_bn.running_var.min()) == 1.0
    # aggregation formula against the oracle's restatement
    sd = synthetic.procedural_state_dict(xo.state_template("M", 400, 8), 3)
    net.load_state_dict(sd)
    assert net.aggregate_sub_bn_stats() == 84
    agg = xo.aggregate_sub_bn(sd, 8)
    got = net.state_dict()
    for k, v in agg.items():
        assert torch.allclose(got[k], v, rtol=1e-6, atol=1e-7), k


def test_cpu_input_fails_loudly():
    net = x3d.generate_model("M", base_bn_splits=1)
    with pytest.raises(_lib.X3DHipError):
        net(torch.zeros(1, 3, 4, 32, 32))


def test_init_statistics():
    torch.manual_seed(0)
    net = x3d.generate_model("M")
    w = net.layer3[0].conv1.weight          # [216, 96, 1,1,1], fan_out = 216
    assert abs(w.std().item() - (2.0 / 216) ** 0.5) < 0.01
    assert float(net.layer1[0].bn2.weight.detach().min()) == 1.0 and float(net.layer1[0].bn2.bias.detach().abs().max()) == 0.0
