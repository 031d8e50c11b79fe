"""Procedural weights and synthetic clips for parity and throughput runs.

No checkpoint of the reference travels (the four ``models/*.pt`` files are absent,
``/root/reference/.MISSING_LARGE_BLOBS``), so both sides of every parity check
regenerate identical tensors from a (name, seed) keyed counter-based RNG
(numpy Philox).  The generated values follow the reference's init law
(``x3d.py:246-250``: kaiming-normal fan_out on every Conv3d; default
Linear/conv-bias init) but BN affine parameters and running statistics are
perturbed away from (1, 0, 0, 1) so that a kernel which ignored them would fail.

Shapes of synthetic clips follow SURVEY.md section 8(d): clips float32[B,3,T,H,W]
i.i.d. N(0,1), labels int64[B,1] uniform in [0, n_classes).
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    key = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return np.random.Generator(np.random.Philox(key=[key, int(seed) & 0xFFFFFFFF]))


def procedural_state_dict(template, seed=0):
    """Return {name: tensor} with the same keys/shapes/dtypes as ``template``
    (a ``state_dict()`` of any X3D variant), filled deterministically."""
    out = {}
    for name, ref in template.items():
        shape = tuple(ref.shape)
        g = _rng(name, seed)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros((), dtype=ref.dtype)
            continue
        if leaf == "running_mean":
            v = 0.1 * g.standard_normal(shape)
        elif leaf == "running_var":
            v = 1.0 + 0.2 * np.abs(g.standard_normal(shape))
        elif len(shape) == 5:  # Conv3d weight: kaiming normal, fan_out, relu gain
            fan_out = shape[0] * shape[2] * shape[3] * shape[4]
            v = g.standard_normal(shape) * np.sqrt(2.0 / fan_out)
        elif len(shape) == 2:  # Linear weight: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            bound = 1.0 / np.sqrt(shape[1])
            v = g.uniform(-bound, bound, shape)
        elif leaf == "weight":  # BN gamma
            v = 1.0 + 0.1 * g.standard_normal(shape)
        elif leaf == "bias":
            v = 0.1 * g.standard_normal(shape)
        else:
            raise KeyError("unexpected state entry %s %s" % (name, shape))
        out[name] = torch.from_numpy(np.ascontiguousarray(v)).to(ref.dtype)
    return out


def synthetic_clips(B, T, H, W, seed=1234, C=3):
    g = _rng("clips", seed)
    x = g.standard_normal((B, C, T, H, W), dtype=np.float32)
    return torch.from_numpy(x)


def synthetic_labels(B, n_classes=400, seed=1234):
    g = _rng("labels", seed)
    y = g.integers(0, n_classes, size=(B, 1), dtype=np.int64)
    return torch.from_numpy(y)


def gradient_sketch(grads, nproj=16, seed=7):
    """Random-projection sketch of a {name: tensor} gradient set: nproj fixed Gaussian
    directions over the concatenated parameters (Philox keyed by parameter name, so any
    side regenerates the same directions).  ||sketch(a) - sketch(b)|| / ||sketch(b)||
    estimates the relative error of the whole gradient vector without shipping it."""
    acc = np.zeros(nproj, dtype=np.float64)
    for name, g in grads.items():
        if torch.is_tensor(g):
            g = g.detach().double().cpu().numpy()
        v = np.asarray(g, dtype=np.float64).reshape(-1)
        r = _rng("sketch:" + name, seed).standard_normal((nproj, v.size), dtype=np.float32)
        acc += r.astype(np.float64) @ v
    return acc


def synthetic_frames_u8(n, h, w, seed=0):
    """Deterministic synthetic "decoded video": uint8 [n][h][w][3] (low-frequency pattern + noise).  numpy only, so the
    golden generator, the CPU tests and the GPU tests regenerate identical frames from (n, h, w, seed)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((n, h, w, 3), dtype=np.uint8)
    for t in range(n):
        for c in range(3):
            base = 127 + 90 * np.sin(0.07 * xx * (c + 1) + 0.3 * t) * np.cos(0.05 * yy + 0.2 * c)
            out[t, :, :, c] = np.clip(base + rng.normal(0, 25, size=(h, w)), 0, 255).astype(np.uint8)
    return out
