"""CPU restatement of the reference's per-clip input pipeline (TEST INFRASTRUCTURE ONLY -- nothing under
x3d-multigrid_amd/ imports this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may).

Follows, with every random draw made an explicit argument:
  * TemporalRandomCrop.__call__            /root/reference/transforms/temporal_transforms.py:94-117
  * the t_stride / frame-index logic of    /root/reference/kinetics_multigrid.py:240-247
  * MultiScaleRandomCropMultigrid.__call__ /root/reference/transforms/spatial_transforms.py:480-495
      (crop box from (scale, tl_x, tl_y), then PIL `resize((size, size), BILINEAR)`)
  * RandomHorizontalFlip (p < 0.5 flips)   spatial_transforms.py:334-346
  * ToTensor(255) + Normalize(mean, std)   spatial_transforms.py:44-83,106-116
  * clip = stack(frames).permute(1,0,2,3)  kinetics_multigrid.py:249-253  -> float32 [3][T][S][S]

The bilinear resize is PIL's (third-party: Pillow, 12.2.0 in this image; the reference pins none): libImaging
Resample.c `precompute_coeffs` + `normalize_coeffs_8bpc` + the two 8-bit passes (horizontal, then vertical, uint8
intermediate, 22-bit fixed-point coefficients, round-half-up, clip) restated in numpy integers.

Pinned by tests/golden/input_*.npz, produced by tests/golden/make_golden_input.py from the reference's own transform
classes (which call PIL) on synthetic uint8 frames.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
KINETICS_MEAN = [110.63666788 / 255, 103.16065604 / 255, 96.29023126 / 255]      # train_x3d_kinetics_multigrid.py:45
KINETICS_STD = [38.7568578 / 255, 37.88248729 / 255, 40.02898126 / 255]          # :46


def temporal_random_crop(frame_indices, begin_index, t_stride, size, gamma_tau, trunc=None):
    """temporal_transforms.py:94-117 with the random begin_index given (reference draws
    random.randint(0, max(0, len - size - 1)))."""
    trunc = size if trunc is None else trunc
    end_index = min(begin_index + size, len(frame_indices))
    out = list(frame_indices[begin_index:end_index:t_stride * gamma_tau])
    out = out[:trunc // gamma_tau]
    i = 0                       # the reference appends to the list it is iterating over: the walk continues into the copies
    while i < len(out):
        if len(out) >= trunc // gamma_tau:
            break
        out.append(out[i])
        i += 1
    return out


def crop_box(width, height, scale, tl_x, tl_y):
    """spatial_transforms.py:482-491."""
    min_length = min(width, height)
    crop_size = int(min_length * scale)
    x1 = int(tl_x * (width - crop_size))
    y1 = int(tl_y * (height - crop_size))
    return x1, y1, crop_size


def resize_coeffs(in_size, out_size):
    """Pillow Resample.c precompute_coeffs (triangle filter, support 1, box = whole image) + normalize_coeffs_8bpc.
    Returns (kk int32 [out][ksize], bounds int32 [out][2] = (first input index, tap count))."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            w = 1.0 - abs(a) if abs(a) < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(v - 0.5) if v < 0 else int(v + 0.5)
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_size):
    """img uint8 [H][W][C] -> uint8 [out][out][C], bit-exact with PIL Image.resize(..., BILINEAR)."""
    H, W, C = img.shape
    kh, bh = resize_coeffs(W, out_size)
    kv, bv = resize_coeffs(H, out_size)
    tmp = np.zeros((H, out_size, C), dtype=np.uint8)
    for xx in range(out_size):
        x0, n = bh[xx]
        acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for i in range(n):
            acc += img[:, x0 + i, :].astype(np.int64) * int(kh[xx, i])
        tmp[:, xx, :] = _clip8(acc)
    out = np.zeros((out_size, out_size, C), dtype=np.uint8)
    for yy in range(out_size):
        y0, n = bv[yy]
        acc = np.full((out_size, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for i in range(n):
            acc += tmp[y0 + i].astype(np.int64) * int(kv[yy, i])
        out[yy] = _clip8(acc)
    return out


def spatial(frame, x1, y1, crop_size, out_size, flip, mean=KINETICS_MEAN, std=KINETICS_STD):
    """One frame uint8 [H][W][3] -> float32 [3][S][S]: crop, PIL bilinear resize, horizontal flip, /255, normalise."""
    img = resize_bilinear_u8(frame[y1:y1 + crop_size, x1:x1 + crop_size, :], out_size)
    if flip:
        img = img[:, ::-1, :]
    t = img.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    for c in range(3):
        t[c] = (t[c] - np.float32(mean[c])) / np.float32(std[c])
    return t


def clip(frames, frame_idx, x1, y1, crop_size, out_size, flip, mean=KINETICS_MEAN, std=KINETICS_STD):
    """frames uint8 [Tsrc][H][W][3]; frame_idx = 0-based positions into frames -> float32 [3][T][S][S]."""
    return np.stack([spatial(frames[i], x1, y1, crop_size, out_size, flip, mean, std) for i in frame_idx], axis=1)


def center_crop_box(width, height):
    """CenterCropScaled.__call__ (spatial_transforms.py:214-228): the centred square of side min(w, h)."""
    crop_size = min(width, height)
    x1 = int(round((width - crop_size) / 2.))
    y1 = int(round((height - crop_size) / 2.))
    return x1, y1, crop_size


def val_crop_indices(n_frames, gamma_tau, sample_duration, crops):
    """kinetics.py:214-233 (the validation dataset): every gamma_tau-th frame of the video, `crops` temporal windows
    of sample_duration // gamma_tau frames starting at 0, step, 2*step, ...  Returns `crops` lists of 0-based frames."""
    strided = list(range(0, n_frames))[::gamma_tau]
    frames = sample_duration // gamma_tau
    step = int((len(strided) - 1 - frames) // (crops - 1))
    if step == 0:
        starts = [0] * crops
    else:
        starts = list(range(0, step * crops, step))
    return [strided[s0:s0 + frames] for s0 in starts]


def val_clips(frames, gamma_tau, sample_duration, crops, out_size, mean=KINETICS_MEAN, std=KINETICS_STD):
    """frames uint8 [n][H][W][3] -> float32 [crops][3][T][S][S] (kinetics.py:205-239 with the validation transforms of
    train_x3d_kinetics_multigrid.py:132-136)."""
    n, h, w, _ = frames.shape
    x1, y1, crop = center_crop_box(w, h)
    return np.stack([clip(frames, idx, x1, y1, crop, out_size, False, mean, std)
                     for idx in val_crop_indices(n, gamma_tau, sample_duration, crops)], axis=0)
