"""Data-parallel path on CPU (gloo, world_size 2): the product's GradReducer sums per-rank flat
gradients exactly like the single-process large batch would produce them.

Equivalence checked: rank r holds samples {r, r+2} of a 4-clip batch and
normalises them with its own BN statistics (1 split); the average of the two ranks' gradients of
their local mean losses equals the gradient of ONE process running all 4 clips with
num_splits=2 (split j = samples j, j+2: the reference's interleaved split mapping, x3d.py:50)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import x3d_oracle as xo
from x3dhip import synthetic
from x3dhip.trainer import GradReducer


def _f64(sd):
    # fp64 so that the comparison tests the algebra, not fp32 ReLU-flip noise (tests/parity.py)
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd = _f64(synthetic.procedural_state_dict(xo.state_template("M", 400, 1), 0))
    x = synthetic.synthetic_clips(4, 2, 32, 32).double()
    y = synthetic.synthetic_labels(4)
    xs, ys = x[rank::world].contiguous(), y[rank::world].contiguous()
    _, loss, grads, _ = xo.train_step_grads(xs, ys, sd, "M", 1)
    names = list(grads.keys())
    flat = torch.cat([grads[k].reshape(-1) for k in names]).contiguous()
    n = flat.numel()
    red = GradReducer(flat, [(n // 2, n), (0, n // 2)], world, None)     # two buckets, late layers first
    red.reduce()
    flat /= world
    if rank == 0:
        np.save(os.path.join(out_dir, "ddp_flat.npy"), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process_large_batch(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "ddp_flat.npy"))
    sd = _f64(synthetic.procedural_state_dict(xo.state_template("M", 400, 2), 0))
    # same parameters; split-BN buffers differ in shape only
    x = synthetic.synthetic_clips(4, 2, 32, 32).double()
    y = synthetic.synthetic_labels(4)
    _, _, grads, _ = xo.train_step_grads(x, y, sd, "M", 2)
    ref = torch.cat([g.reshape(-1) for g in grads.values()]).numpy()
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel < 1e-9, rel      # fp64 on both sides: the equivalence is exact up to summation order


def test_bucket_order_is_head_first():
    import x3d
    from x3dhip.trainer import FlatParams
    net = x3d.generate_model("M", base_bn_splits=1)
    fp = FlatParams.__new__(FlatParams)
    # offsets without touching a GPU
    fp.offsets, o = [], 0
    for _, p in net.named_parameters():
        fp.offsets.append((o, p.numel()))
        o += p.numel()
    fp.numel = o
    b = fp.head_first_buckets(net)
    assert b[0][1] == 3794322 and b[1][0] == 0 and b[0][0] == b[1][1]
    names = [k for k, _ in net.named_parameters()]
    first_l3 = fp.offsets[[i for i, k in enumerate(names) if k.startswith("layer3.")][0]][0]
    assert b[0][0] == first_l3
