"""Input-pipeline oracle (oracle/input_oracle.py) against the goldens produced by the reference's own transform
classes (tests/golden/make_golden_input.py): frame indices and the float clip, bit-exact."""
import glob
import os

import numpy as np
import pytest

from oracle import input_oracle as io
from x3dhip.synthetic import synthetic_frames_u8

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "input_*.npz")))


def load_case(path):
    g = np.load(path)
    frames = synthetic_frames_u8(int(g["n_frames"]), int(g["h"]), int(g["w"]), int(g["frames_seed"]))
    return g, frames


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[6:-4] for p in GOLD])
def test_oracle_matches_reference_transforms(path):
    g, frames = load_case(path)
    idx = io.temporal_random_crop(list(range(1, int(g["n_frames"]) + 1)), int(g["begin_index"]), int(g["t_stride"]),
                                  int(g["num_frames"]), int(g["gamma_tau"]), trunc=int(g["sample_duration"]))
    assert idx == [int(v) for v in g["frame_idx"]]
    x1, y1, crop = io.crop_box(int(g["w"]), int(g["h"]), float(g["scale"]), float(g["tl_x"]), float(g["tl_y"]))
    clip = io.clip(frames, [i - 1 for i in idx], x1, y1, crop, int(g["c_size"]), float(g["p"]) < 0.5)
    assert clip.shape == g["clip"].shape
    assert np.array_equal(clip, g["clip"]), "max abs diff %g" % np.abs(clip - g["clip"]).max()


def test_resize_restatement_is_pil_exact():
    """the numpy restatement of Pillow's bilinear resample against Pillow itself (present in this image)"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    for crop, out in [(256, 224), (200, 112), (137, 158), (64, 224), (224, 224), (300, 79)]:
        a = rng.integers(0, 256, size=(crop, crop, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(a).resize((out, out), Image.BILINEAR))
        assert np.array_equal(io.resize_bilinear_u8(a, out), ref), (crop, out)


VGOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "inputval_*.npz")))


@pytest.mark.parametrize("path", VGOLD, ids=[os.path.basename(p)[9:-4] for p in VGOLD])
def test_validation_clips_match_reference_transforms(path):
    g = np.load(path)
    frames = synthetic_frames_u8(int(g["n_frames"]), int(g["h"]), int(g["w"]), int(g["frames_seed"]))
    clips = io.val_clips(frames, int(g["gamma_tau"]), int(g["sample_duration"]), int(g["crops"]), int(g["c_size"]))
    assert clips.shape == g["clips"].shape
    assert np.array_equal(clips, g["clips"])
