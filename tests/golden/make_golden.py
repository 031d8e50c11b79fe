"""Generate golden vectors by running the REFERENCE's own x3d.py / cycle_batch_sampler.py.

Run in the build container only (the reference checkout never travels to the GPU box):

    python tests/golden/make_golden.py [--only NAME] [--big]

It imports /root/reference/x3d.py and /root/reference/cycle_batch_sampler.py by path
(both depend on torch only), loads procedural weights (x3dhip/synthetic.py), feeds
synthetic clips and stores inputs' seeds + expected outputs as small .npz fixtures next
to this file.  Fixtures are data only: logits, loss, per-parameter gradient L2 norms,
a few small full gradients, split-BN running statistics after one step, aggregated BN
statistics, and the sampler's batch-length / long-index sequence.
"""
import argparse
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "x3d-multigrid_amd"))
from x3dhip import synthetic  # noqa: E402

REF = "/root/reference"


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


FULL_GRADS = ["conv1_s.weight", "conv1_t.weight", "bn1.weight", "bn1.bias",
              "layer1.0.conv2.weight", "layer1.0.fc1.bias", "layer1.0.fc2.bias",
              "layer1.0.fc1.weight", "layer1.0.bn2.weight", "layer1.0.bn2.bias",
              "layer1.0.downsample.1.weight", "layer2.1.conv2.weight",
              "layer3.4.fc2.weight", "layer4.6.bn3.bias", "layer4.6.conv2.weight", "fc2.bias"]
FULL_STATS = ["bn1", "layer1.0.bn2", "layer1.0.downsample.1", "layer2.3.bn1", "layer3.10.bn3",
              "layer4.6.bn2", "bn5"]


def _grad_record(net, out, tag):
    names, norms, grads = [], [], {}
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(p.grad.double().norm().item())
        grads[k] = p.grad
    out["grad_names"] = np.array(names)
    out["grad_norms" + tag] = np.array(norms)
    out["grad_global_norm" + tag] = np.float64(np.sqrt(np.sum(np.square(norms))))
    out["grad_sketch" + tag] = synthetic.gradient_sketch(grads)
    for k in FULL_GRADS:
        out["grad%s/%s" % (tag, k)] = grads[k].numpy()
    return grads


def _ref_model(ref, version, **kw):
    """The reference's generate_model (x3d.py:365-374).  'L' is not a key of the reference's tables (x3d.py:352-363):
    SURVEY.md section 7 step 10 defines it as the XL depth at M widths, which the reference's own ResNet / Bottleneck
    classes build when given those two table rows."""
    if version == "L":
        return ref.ResNet(ref.Bottleneck, ref.get_blocks("XL"), ref.get_inplanes("M"), **kw)
    return ref.generate_model(version, **kw)


def train_case(ref, version, B, T, H, splits, seed=0, task="class", second_draw_threads=0):
    """One training step of the reference in fp32 (the parity target) and in fp64 (the
    same reference code after .double(): the exact-arithmetic value, which measures how
    much of an fp32 discrepancy is the reference's own rounding noise)."""
    x = synthetic.synthetic_clips(B, T, H, H, seed=1234)
    y = synthetic.synthetic_labels(B, seed=1234)
    out = {"shape": np.array([B, T, H, splits]), "seed": np.array([seed, 1234])}
    g32 = None
    for tag, dt in (("", torch.float32), ("64", torch.float64)):
        torch.manual_seed(0)
        net = _ref_model(ref, version, n_classes=400, dropout=0.0, base_bn_splits=splits, task=task)
        sd = synthetic.procedural_state_dict(net.state_dict(), seed)
        net.load_state_dict(sd)
        net = net.to(dt)
        net.train(True)
        logits = net(x.to(dt))
        # task='loc' (x3d.py:340-343): per-frame logits [B, C, T]; the same label on every frame gives a scalar loss
        # that exercises the whole per-frame head (the Charades losses of train_x3d_charades_loc.py are out of scope)
        loss = torch.nn.CrossEntropyLoss()(logits, y if task == "class" else y.expand(B, logits.shape[2]))
        loss.backward()
        out["logits" + tag] = logits.detach().numpy()[:, :, 0] if task == "class" else logits.detach().numpy()
        out["loss" + tag] = np.float64(loss.item())
        grads = _grad_record(net, out, tag)
        if tag == "":
            g32 = {k: v.double() for k, v in grads.items()}
        else:
            out["ref32_vs_64_full_rel"] = np.array(
                [((g32[k] - grads[k]).norm() / grads[k].norm().clamp_min(1e-300)).item() for k in grads])
            del g32
        st = net.state_dict()
        if tag == "":
            rm_names, rm_norm, rv_norm = [], [], []
            for k, v in st.items():
                if k.endswith(".split_bn.running_mean"):
                    p = k[: -len(".split_bn.running_mean")]
                    rm_names.append(p)
                    rm_norm.append(v.double().norm().item())
                    rv_norm.append(st[p + ".split_bn.running_var"].double().norm().item())
            out["bn_names"] = np.array(rm_names)
            out["bn_rm_norms"] = np.array(rm_norm)
            out["bn_rv_norms"] = np.array(rv_norm)
            for p in FULL_STATS:
                out["rm/" + p] = st[p + ".split_bn.running_mean"].numpy()
                out["rv/" + p] = st[p + ".split_bn.running_var"].numpy()
            # eval after aggregation (x3d.py:306-313), same clip
            net.train(False)
            n_agg = net.aggregate_sub_bn_stats()
            st = net.state_dict()
            for p in FULL_STATS:
                out["agg_rm/" + p] = st[p + ".bn.running_mean"].numpy()
                out["agg_rv/" + p] = st[p + ".bn.running_var"].numpy()
            out["n_agg"] = np.array(n_agg)
            with torch.no_grad():
                out["eval_logits"] = net(x).numpy()[:, :, 0] if task == "class" else net(x).numpy()
        del net, logits, loss, grads
    if second_draw_threads:
        # A second fp32 run of the reference with another intra-op thread count (= another summation order in its
        # convolutions and reductions): an independent draw of the reference's own fp32 rounding noise.  At the 55-block
        # depth one draw under-estimates the floor (global gradient norm vs fp64: 3.0e-4 with 8 threads, 1.5e-3 with 1).
        nt = torch.get_num_threads()
        torch.set_num_threads(second_draw_threads)
        torch.manual_seed(0)
        net = _ref_model(ref, version, n_classes=400, dropout=0.0, base_bn_splits=splits, task=task)
        net.load_state_dict(synthetic.procedural_state_dict(net.state_dict(), seed))
        net.train(True)
        logits = net(x)
        loss = torch.nn.CrossEntropyLoss()(logits, y if task == "class" else y.expand(B, logits.shape[2]))
        loss.backward()
        rec = {}
        _grad_record(net, rec, "")
        out["grad_norms_draw2"] = rec["grad_norms"]
        out["grad_global_norm_draw2"] = rec["grad_global_norm"]
        out["grad_sketch_draw2"] = rec["grad_sketch"]
        for k in FULL_GRADS:
            out["grad_draw2/" + k] = rec["grad/" + k]
        torch.set_num_threads(nt)
    return out


def eval_case(ref, version, B, T, H, seed=0):
    torch.manual_seed(0)
    net = ref.generate_model(version, n_classes=400, dropout=0.0, base_bn_splits=1)
    sd = synthetic.procedural_state_dict(net.state_dict(), seed)
    net.load_state_dict(sd)
    net.train(False)
    x = synthetic.synthetic_clips(B, T, H, H, seed=1234)
    with torch.no_grad():
        logits = net(x)
    return {"logits": logits.numpy()[:, :, 0], "shape": np.array([B, T, H, 1]),
            "seed": np.array([seed, 1234])}


def keys_case(ref):
    out = {}
    for v, splits in (("M", 1), ("M", 4), ("XL", 2)):
        net = ref.generate_model(v, n_classes=400, base_bn_splits=splits)
        sd = net.state_dict()
        out["keys_%s_%d" % (v, splits)] = np.array(list(sd.keys()))
        out["shapes_%s_%d" % (v, splits)] = np.array(
            ["x".join(map(str, t.shape)) for t in sd.values()])
        out["params_%s_%d" % (v, splits)] = np.array([k for k, _ in net.named_parameters()])
    # update_bn_splits_long_cycle re-creates split_bn (x3d.py:298-303)
    net = ref.generate_model("M", base_bn_splits=2)
    r = net.update_bn_splits_long_cycle(4)
    out["update_ret"] = np.array(r)
    out["update_shape"] = np.array(net.state_dict()["bn1.split_bn.running_mean"].shape)
    return out


def sampler_case(cbs, batch_size, schedule, cur, n):
    class _Src:  # sampler that yields forever
        def __iter__(self):
            i = 0
            while True:
                yield i
                i += 1

        def __len__(self):
            return 1 << 30
    bs = cbs.CycleBatchSampler(_Src(), batch_size, False, schedule=list(schedule),
                               cur_iterations=cur, long_cycle_bs_scale=[8, 4, 2, 1])
    lens, longs = [], []
    it = iter(bs)
    for _ in range(n):
        b = next(it)
        lens.append(len(b))
        longs.append(b[0][1])
    return np.array(lens), np.array(longs)


def sampler_cases(cbs):
    out = {}
    sch = [int(f * 120 * 1718) for f in (0, 0.4, 0.65, 0.85, 1)]
    out["schedule_full"] = np.array(sch)
    l, g = sampler_case(cbs, 128, sch, 0, 12)
    out["full_first_len"], out["full_first_long"] = l, g
    l, g = sampler_case(cbs, 128, sch, 204000, 12)
    out["full_resume_len"], out["full_resume_long"] = l, g
    # compact schedule: every transition inside 400 iterations, per-GPU base B=8
    small = [0, 160, 260, 340, 400]
    out["schedule_small"] = np.array(small)
    l, g = sampler_case(cbs, 8, small, 0, 400)
    out["small_len"], out["small_long"] = l, g
    l, g = sampler_case(cbs, 8, small, 200, 150)
    out["small_resume_len"], out["small_resume_long"] = l, g
    # transition points of the full schedule, found by bisection-free scan of long idx
    # (cheap: iterate with batch_size=1 so each batch is tiny)
    l, g = sampler_case(cbs, 1, sch, 0, 206100)
    chg = np.nonzero(np.diff(g))[0] + 1
    out["full_transitions_at"] = chg
    out["full_transitions_to"] = g[chg]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--big", action="store_true", help="also (re)generate the config-2 case")
    args = ap.parse_args()
    warnings.filterwarnings("ignore")
    torch.set_num_threads(8)
    ref = _load("x3d")
    cbs = _load("cycle_batch_sampler")
    jobs = {
        "keys": lambda: keys_case(ref),
        "sampler": lambda: sampler_cases(cbs),
        "eval_S_2x13x160": lambda: eval_case(ref, "S", 2, 13, 160),
        "train_M_2x4x32_s1": lambda: train_case(ref, "M", 2, 4, 32, 1),
        "train_M_8x4x64_s2": lambda: train_case(ref, "M", 8, 4, 64, 2),
        "train_M_16x2x47_s4": lambda: train_case(ref, "M", 16, 2, 47, 4),
        "train_M_2x4x111_s1": lambda: train_case(ref, "M", 2, 4, 111, 1),
        "train_M_2x4x158_s2": lambda: train_case(ref, "M", 2, 4, 158, 2),
        "train_M_2x8x112_s1": lambda: train_case(ref, "M", 2, 8, 112, 1),
        "trainloc_M_2x4x64_s1": lambda: train_case(ref, "M", 2, 4, 64, 1, task="loc"),
    }
    if args.big:
        jobs["train_M_8x16x224_s1"] = lambda: train_case(ref, "M", 8, 16, 224, 1)
        # BASELINE config 3 at full per-GPU batch (kinetics_multigrid.py:205-237 x cycle_batch_sampler.py:98-111):
        # the two literal shapes of the config with num_splits = B / 8
        jobs["train_M_64x4x112_s8"] = lambda: train_case(ref, "M", 64, 4, 112, 8)
        jobs["train_M_16x16x224_s2"] = lambda: train_case(ref, "M", 16, 16, 224, 2)
        # the largest-N shape of the reference's own shape table (long cycle 0, short-cycle step 0): B = 128 per GPU,
        # T = 4, odd 111 x 111 crop, 8 BN splits of 16 samples
        jobs["train_M_128x4x111_s8"] = lambda: train_case(ref, "M", 128, 4, 111, 8)
    if args.big:
        # BASELINE config 5's literal clip shape (T = 16, H = W = 312) on the "L" architecture, B = 2
        jobs["train_L_2x16x312_s1"] = lambda: train_case(ref, "L", 2, 16, 312, 1, seed=4, second_draw_threads=1)
    # X3D-XL widths (x3d.py:355) pinned by the reference itself on a tiny clip
    jobs["train_XL_2x4x64_s1"] = lambda: train_case(ref, "XL", 2, 4, 64, 1, seed=2)
    # "X3D-L" (BASELINE config 5's architecture: XL depth, M widths) in fp32 -- the target of the mixed-storage (bf16) mode
    jobs["train_L_4x4x96_s1"] = lambda: train_case(ref, "L", 4, 4, 96, 1, seed=3, second_draw_threads=1)
    for name, fn in jobs.items():
        if args.only and args.only != name:
            continue
        res = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
