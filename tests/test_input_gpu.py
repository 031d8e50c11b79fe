"""GPU clip input pipeline (csrc/clip.hip through the C ABI) against the goldens of the reference's transforms and
against the oracle on further shapes: bit-exact (uint8 resample) and exact fp32 normalisation."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import input_oracle as io
from x3dhip.synthetic import synthetic_frames_u8

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "input_*.npz")))


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[6:-4] for p in GOLD])
def test_golden_clips(path):
    from x3dhip import clip_input as ci
    g = np.load(path)
    frames = synthetic_frames_u8(int(g["n_frames"]), int(g["h"]), int(g["w"]), int(g["frames_seed"]))
    x1, y1, crop = io.crop_box(int(g["w"]), int(g["h"]), float(g["scale"]), float(g["tl_x"]), float(g["tl_y"]))
    p = dict(frame_idx=[int(v) - 1 for v in g["frame_idx"]], x1=x1, y1=y1, crop=crop, out=int(g["c_size"]),
             flip=float(g["p"]) < 0.5)
    pre = ci.ClipPreprocessor(_dev())
    out = pre([(torch.from_numpy(frames).to(_dev()), p)])
    torch.cuda.synchronize()
    got = out[0].cpu().numpy()
    assert got.shape == g["clip"].shape
    assert np.array_equal(got, g["clip"]), "max abs diff %g" % np.abs(got - g["clip"]).max()


def test_batch_of_mixed_crops_matches_oracle():
    """one launch, 5 samples with different source sizes / crops / flips (incl. upscaling and a 1-pixel-margin crop)"""
    from x3dhip import clip_input as ci
    dev = _dev()
    rng = np.random.default_rng(5)
    specs = [(20, 90, 120, 64, 10, 3, 1), (12, 128, 128, 128, 0, 0, 0), (16, 70, 200, 33, 150, 30, 1),
             (9, 300, 260, 257, 2, 40, 0), (30, 64, 64, 17, 40, 41, 1)]
    T, S = 6, 40
    samples, expect = [], []
    for (n, h, w, crop, x1, y1, flip) in specs:
        fr = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
        idx = [int(v) for v in rng.integers(0, n, size=T)]
        samples.append((torch.from_numpy(fr).to(dev), dict(frame_idx=idx, x1=x1, y1=y1, crop=crop, out=S, flip=bool(flip))))
        expect.append(io.clip(fr, idx, x1, y1, crop, S, bool(flip)))
    out = ci.ClipPreprocessor(dev)(samples)
    torch.cuda.synchronize()
    for b, e in enumerate(expect):
        assert np.array_equal(out[b].cpu().numpy(), e), (b, np.abs(out[b].cpu().numpy() - e).max())


def test_rejects_bad_boxes():
    from x3dhip import clip_input as ci
    dev = _dev()
    fr = torch.zeros((4, 32, 32, 3), dtype=torch.uint8, device=dev)
    pre = ci.ClipPreprocessor(dev)
    with pytest.raises(ValueError):
        pre([(fr, dict(frame_idx=[0, 1], x1=20, y1=0, crop=20, out=8, flip=False))])
    with pytest.raises(ValueError):
        pre([(fr, dict(frame_idx=[0, 4], x1=0, y1=0, crop=20, out=8, flip=False))])


def test_device_video_dataset_protocol():
    """DeviceVideoKinetics.batch: Kinetics.__getitem__'s shape protocol + draws, clips equal to the oracle fed with
    the same draws (random.Random seeded twice)."""
    import random
    from kinetics_multigrid import DeviceVideoKinetics
    from x3dhip import clip_input as ci
    from cycle_batch_sampler import step_clip_shape
    dev = _dev()
    vids_np = [synthetic_frames_u8(n, h, w, seed) for (n, h, w, seed) in [(90, 64, 96, 11), (100, 80, 80, 12), (40, 72, 60, 13)]]
    vids = [torch.from_numpy(v).to(dev) for v in vids_np]
    ds = DeviceVideoKinetics(vids, [3, 7, 9], sample_duration=80, gamma_tau=5, crop_size=64, rng=random.Random(123))
    ref_rng = random.Random(123)
    for (iteration, lc) in [(0, 0), (1, 2), (5, 3)]:
        clips, y, lcs, stats = ds.batch([2, 0, 1], iteration, lc)
        T, S = step_clip_shape(lc, iteration, 80, 5, 64)
        assert tuple(clips.shape) == (3, 3, T, S, S) and y.view(-1).tolist() == [9, 3, 7] and lcs == lc
        frames = ds.long_cycles[lc][0]
        for b, i in enumerate([2, 0, 1]):
            v = vids_np[i]
            p = ci.draw_clip_params(v.shape[0], v.shape[2], v.shape[1], ds.scales, S, 80, 5, frames, rng=ref_rng)
            e = io.clip(v, p["frame_idx"], p["x1"], p["y1"], p["crop"], S, p["flip"])
            assert np.array_equal(clips[b].cpu().numpy(), e)


def test_input_throughput_report(capsys):
    """not a pass/fail bar: prints the clip rate of the base shape (8 x 16 x 224^2 from 256 x 340 sources)"""
    import random
    import time
    from kinetics_multigrid import DeviceVideoKinetics
    dev = _dev()
    vids = [torch.randint(0, 256, (120, 256, 340, 3), dtype=torch.uint8, device=dev) for _ in range(8)]
    ds = DeviceVideoKinetics(vids, list(range(8)), rng=random.Random(0))
    out = torch.empty((8, 3, 16, 224, 224), device=dev)
    for _ in range(3):
        ds.batch(list(range(8)), 2, 3, out=out)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 20
    for _ in range(n):
        ds.batch(list(range(8)), 2, 3, out=out)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    with capsys.disabled():
        print("\n[input pipeline] B=8 T=16 224^2 from 256x340 uint8: %.2f ms/batch = %.0f clips/s (host draws included)" % (dt * 1e3, 8 / dt))


VGOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "inputval_*.npz")))


@pytest.mark.parametrize("path", VGOLD, ids=[os.path.basename(p)[9:-4] for p in VGOLD])
def test_validation_batch_matches_golden(path):
    """DeviceVideoKinetics.val_batch: centre crop scaled + 3 temporal windows (kinetics.py:205-239) on the GPU"""
    from kinetics_multigrid import DeviceVideoKinetics
    dev = _dev()
    g = np.load(path)
    frames = synthetic_frames_u8(int(g["n_frames"]), int(g["h"]), int(g["w"]), int(g["frames_seed"]))
    ds = DeviceVideoKinetics([torch.from_numpy(frames).to(dev)], [5], sample_duration=int(g["sample_duration"]),
                             gamma_tau=int(g["gamma_tau"]), crop_size=int(g["c_size"]))
    clips, y = ds.val_batch([0], crops=int(g["crops"]))
    torch.cuda.synchronize()
    assert tuple(clips.shape) == (1,) + g["clips"].shape and y.tolist() == [5]
    assert np.array_equal(clips[0].cpu().numpy(), g["clips"])
