import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
import torch
from oracle import x3d_oracle as xo
from x3dhip import synthetic, engine
import x3d
B, T, H, S = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda:0")
sd = synthetic.procedural_state_dict(xo.state_template("M", 400, S), 0)
net = x3d.generate_model("M", dropout=0.0, base_bn_splits=S); net.load_state_dict(sd); net.to(dev).train(True)
x = synthetic.synthetic_clips(B, T, H, H)
taps = {}
with torch.no_grad():
    xo.forward(x, sd, "M", S, True, None, taps)
ctx = engine.TrunkContext()
with torch.no_grad():
    engine.trunk_forward(net, x.to(dev), True, ctx)
st = ctx.stem
stem_h = torch.relu(st["c0"][..., 0, None, None, None] * st["a_t"] + st["c0"][..., 1, None, None, None])
print("stem", ((stem_h.cpu() - taps["stem"]).norm() / taps["stem"].norm()).item())
for rec, (name, ref) in zip(ctx.blocks, list(taps.items())[1:]):
    e = ((rec["out"].cpu() - ref).norm() / ref.norm()).item()
    print(name, "%.3e" % e, tuple(ref.shape))
