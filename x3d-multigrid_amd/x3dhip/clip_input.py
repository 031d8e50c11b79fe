"""Host side of the GPU clip input pipeline (csrc/clip.hip): the integer/random logic of the reference's
per-sample transforms and the job tables for x3d_clip_preprocess.

Reference call sites mirrored (random draws in the reference's order, from Python's `random` like the reference):
  kinetics_multigrid.py:240-253 (__getitem__), transforms/temporal_transforms.py:94-117 (TemporalRandomCrop),
  transforms/spatial_transforms.py:480-503 (MultiScaleRandomCropMultigrid), :334-349 (RandomHorizontalFlip),
  :44-83,106-116 (ToTensor(255), Normalize).  The arithmetic runs on the GPU; nothing here falls back to CPU.
"""
import math
import random as _random

import numpy as np
import torch

from . import _lib

PRECISION_BITS = 32 - 8 - 2
KINETICS_MEAN = [110.63666788 / 255, 103.16065604 / 255, 96.29023126 / 255]      # train_x3d_kinetics_multigrid.py:45
KINETICS_STD = [38.7568578 / 255, 37.88248729 / 255, 40.02898126 / 255]          # :46


def temporal_random_crop(frame_indices, size, gamma_tau, t_stride=1, trunc=None, rng=_random):
    """TemporalRandomCrop.__call__ (temporal_transforms.py:94-117), including its looping of short videos."""
    trunc = size if trunc is None else trunc
    rand_end = max(0, len(frame_indices) - size - 1)
    begin_index = rng.randint(0, rand_end)
    end_index = min(begin_index + size, len(frame_indices))
    out = list(frame_indices[begin_index:end_index:t_stride * gamma_tau])
    out = out[:trunc // gamma_tau]
    i = 0
    while i < len(out):                     # the reference appends to the list it iterates over
        if len(out) >= trunc // gamma_tau:
            break
        out.append(out[i])
        i += 1
    return out


def draw_clip_params(n_frames, width, height, scales, c_size, num_frames, gamma_tau, sample_duration, rng=_random):
    """All random draws of one __getitem__ in the reference's order (kinetics_multigrid.py:245-251):
    t_stride, TemporalRandomCrop's begin index, then Compose.randomize_parameters (scale, tl_x, tl_y, flip p)."""
    t_stride = rng.randint(1, max(1, num_frames // sample_duration))
    idx = temporal_random_crop(list(range(1, n_frames + 1)), num_frames, gamma_tau, t_stride, sample_duration, rng)
    scale = scales[rng.randint(0, len(scales) - 1)]
    tl_x = rng.random()
    tl_y = rng.random()
    p = rng.random()
    min_length = min(width, height)
    crop = int(min_length * scale)
    x1 = int(tl_x * (width - crop))
    y1 = int(tl_y * (height - crop))
    return dict(frame_idx=[i - 1 for i in idx], x1=x1, y1=y1, crop=crop, out=c_size, flip=p < 0.5, t_stride=t_stride,
                scale=scale, tl_x=tl_x, tl_y=tl_y, p=p)


def center_crop_box(width, height):
    """CenterCropScaled (spatial_transforms.py:214-228): centred square of side min(w, h); Python's round (half to even)
    like the reference."""
    crop = min(width, height)
    return int(round((width - crop) / 2.)), int(round((height - crop) / 2.)), crop


def val_crop_indices(n_frames, gamma_tau, sample_duration, crops):
    """Validation windows of kinetics.py:214-233: every gamma_tau-th frame, `crops` windows of sample_duration //
    gamma_tau frames at 0, step, 2*step (step 0 -> identical windows).  0-based frame lists."""
    strided = list(range(n_frames))[::gamma_tau]
    frames = sample_duration // gamma_tau
    step = int((len(strided) - 1 - frames) // (crops - 1))
    if step < 0 or len(strided) < frames:
        raise ValueError("video of %d frames is too short for %d-frame validation windows at stride %d" %
                         (n_frames, frames, gamma_tau))
    starts = [0] * crops if step == 0 else list(range(0, step * crops, step))
    return [strided[s0:s0 + frames] for s0 in starts]


_coeff_cache = {}


def resize_coeffs(in_size, out_size):
    """Pillow's bilinear coefficient table for in_size -> out_size (Resample.c precompute_coeffs with the triangle
    filter + normalize_coeffs_8bpc), vectorised; double precision in the same operation order as the C code."""
    key = (in_size, out_size)
    hit = _coeff_cache.get(key)
    if hit is not None:
        return hit
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # (int) truncation of a positive value
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    a = (x + xmin[:, None] - center[:, None] + 0.5) * ss
    w = np.where(np.abs(a) < 1.0, 1.0 - np.abs(a), 0.0)
    w = np.where(x < n[:, None], w, 0.0)
    ww = np.zeros(out_size)
    for i in range(ksize):                   # left-to-right accumulation like the C loop
        ww = ww + w[:, i]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    v = k * float(1 << PRECISION_BITS)
    kk = np.where(v < 0, (v - 0.5).astype(np.int64), (v + 0.5).astype(np.int64)).astype(np.int32)
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    out = (np.ascontiguousarray(kk), np.ascontiguousarray(bounds), ksize)
    _coeff_cache[key] = out
    return out


_JOB_DT = np.dtype([("src", "<u8"), ("tmp", "<u8"), ("dst", "<u8"), ("kk", "<u8"), ("bounds", "<u8"), ("frames", "<u8"),
                    ("Hs", "<i4"), ("Ws", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("crop", "<i4"), ("out", "<i4"),
                    ("ksize", "<i4"), ("T", "<i4"), ("flip", "<i4"), ("pad", "<i4")])


class ClipPreprocessor:
    """Turns decoded uint8 videos resident on the GPU into the normalised float NCTHW batch of one training step.

    samples: list of (frames uint8 CUDA tensor [Tsrc, H, W, 3], params dict from draw_clip_params); all samples of
    a step share T and the output size (the multigrid schedule fixes both per step)."""

    def __init__(self, device, mean=KINETICS_MEAN, std=KINETICS_STD):
        self.device = torch.device(device)
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)
        self._tables = {}
        assert _lib.lib().x3d_clip_job_bytes() == _JOB_DT.itemsize

    def _table(self, crop, out):
        key = (crop, out)
        hit = self._tables.get(key)
        if hit is None:
            kk, bounds, ksize = resize_coeffs(crop, out)
            hit = (torch.from_numpy(kk).to(self.device), torch.from_numpy(bounds).to(self.device), ksize)
            self._tables[key] = hit
        return hit

    def __call__(self, samples, out=None):
        L = _lib.lib()
        B = len(samples)
        T = len(samples[0][1]["frame_idx"])
        S = samples[0][1]["out"]
        batch = out if out is not None else torch.empty((B, 3, T, S, S), dtype=torch.float32, device=self.device)
        jobs = np.zeros(B, dtype=_JOB_DT)
        keep = []
        max_crop = 0
        for b, (frames, p) in enumerate(samples):
            if frames.device != self.device or frames.dtype != torch.uint8 or not frames.is_contiguous():
                raise ValueError("frames must be contiguous uint8 tensors on %s" % self.device)
            if len(p["frame_idx"]) != T or p["out"] != S:
                raise ValueError("all samples of a step share T and the output size")
            Tsrc, Hs, Ws, C = frames.shape
            if C != 3 or p["crop"] <= 0 or p["x1"] < 0 or p["y1"] < 0 or p["x1"] + p["crop"] > Ws or p["y1"] + p["crop"] > Hs:
                raise ValueError("crop box outside the frame")
            if min(p["frame_idx"]) < 0 or max(p["frame_idx"]) >= Tsrc:
                raise ValueError("frame index outside the video")
            kk, bounds, ksize = self._table(p["crop"], S)
            fidx = torch.tensor(p["frame_idx"], dtype=torch.int32, device=self.device)
            tmp = torch.empty((T, p["crop"], S, 3), dtype=torch.uint8, device=self.device)
            keep += [fidx, tmp]
            jobs[b] = (frames.data_ptr(), tmp.data_ptr(), batch[b].data_ptr(), kk.data_ptr(), bounds.data_ptr(),
                       fidx.data_ptr(), Hs, Ws, p["x1"], p["y1"], p["crop"], S, ksize, T, 1 if p["flip"] else 0, 0)
            max_crop = max(max_crop, p["crop"])
        jd = torch.from_numpy(jobs.view(np.uint8).copy()).to(self.device)
        _lib.check(L.x3d_clip_preprocess(jd.data_ptr(), B, T, max_crop, S, self.mean.ctypes.data, self.std.ctypes.data,
                                         _lib.stream()))
        for t in keep + [jd]:
            t.record_stream(torch.cuda.current_stream())
        return batch
