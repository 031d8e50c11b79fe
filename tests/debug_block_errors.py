"""Per-block forward / input-gradient error of the HIP schedule against the fp64 oracle (the quantities asserted by
tests/test_model_gpu.py::test_blocks_tight_vs_oracle), printed instead of asserted.  Debug aid for tolerance questions."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import parity  # noqa: E402
from oracle import x3d_oracle as xo  # noqa: E402
from x3dhip import engine, synthetic  # noqa: E402
from test_model_gpu import _build  # noqa: E402

dev = torch.device("cuda:0")
for shape in [(4, 4, 40, 2), (2, 2, 31, 1), (8, 4, 18, 4)]:
    B, T, H, S = shape
    net = _build("M", S, dev)
    net.train(True)
    sd64 = {k: (v.double() if v.is_floating_point() else v)
            for k, v in synthetic.procedural_state_dict(xo.state_template("M", 400, S), 0).items()}
    rows = {r[0]: r for r in xo.block_table("M")}
    g = torch.Generator().manual_seed(5)
    for name in ["layer1.0", "layer1.1", "layer1.2", "layer2.0", "layer3.3", "layer4.0", "layer4.2"]:
        p, cin, cm, co, stride, se, ds = rows[name]
        li, bi = name.split(".")
        blk = getattr(net, li)[int(bi)]
        x = torch.relu(torch.randn(B, cin, T, H, H, generator=g, dtype=torch.float64))
        full = dict(sd64)
        leaf = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if k.startswith(p + ".") and xo.is_parameter(k)}
        full.update(leaf)
        xr = x.clone().requires_grad_(True)
        out_ref = xo.bottleneck(xr, full, p, stride, se, ds, S, True, None)
        dout = torch.randn(out_ref.shape, generator=g, dtype=torch.float64)
        out_ref.backward(dout)
        ctx = engine.TrunkContext()
        packs = engine.weight_packs(net)
        packs.refresh()
        out, _ = engine._block_forward(blk, x.float().to(dev), None, S, True, ctx, packs)
        sink = engine._GradSink(False)
        dprev, _ = engine._block_backward(ctx.blocks[0], dout.float().to(dev), sink)
        sink.flush()
        print(shape, name, "fwd %.2e" % parity.rel(out.cpu().numpy(), out_ref.detach().numpy()),
              "dprev %.2e" % parity.rel(dprev.cpu().numpy(), xr.grad.numpy()))
