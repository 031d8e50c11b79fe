"""Host logic of the GPU input pipeline (x3dhip/clip_input.py), CPU only: the coefficient tables equal the oracle's
(hence Pillow's), and seeding `random` like the golden generator reproduces the reference's draws."""
import glob
import os
import random

import numpy as np
import pytest

from oracle import input_oracle as io
from x3dhip import clip_input as ci

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "input_*.npz")))


@pytest.mark.parametrize("sizes", [(256, 224), (200, 112), (137, 158), (320, 111), (64, 224), (224, 224), (500, 79), (57, 56)])
def test_coefficient_tables_match_oracle(sizes):
    kk, bounds, ksize = ci.resize_coeffs(*sizes)
    kk_o, b_o = io.resize_coeffs(*sizes)
    assert ksize == kk_o.shape[1]
    assert np.array_equal(kk, kk_o) and np.array_equal(bounds, b_o)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[6:-4] for p in GOLD])
def test_draws_follow_the_reference_order(path):
    g = np.load(path)
    random.seed(int(g["frames_seed"]))                      # make_golden_input.py seeds random with the case's seed
    resize_size = [256., 320.]
    p = ci.draw_clip_params(int(g["n_frames"]), int(g["w"]), int(g["h"]), [224 / i for i in resize_size], int(g["c_size"]),
                            int(g["num_frames"]), int(g["gamma_tau"]), int(g["sample_duration"]))
    assert [i + 1 for i in p["frame_idx"]] == [int(v) for v in g["frame_idx"]]
    assert p["t_stride"] == int(g["t_stride"])
    assert p["scale"] == float(g["scale"]) and p["tl_x"] == float(g["tl_x"]) and p["tl_y"] == float(g["tl_y"])
    assert p["p"] == float(g["p"])
    assert (p["x1"], p["y1"], p["crop"]) == io.crop_box(int(g["w"]), int(g["h"]), float(g["scale"]), float(g["tl_x"]),
                                                       float(g["tl_y"]))
