"""Synthetic stand-in for the reference's Kinetics multigrid dataset (kinetics_multigrid.py).

The reference dataset decodes JPEG frame folders on NFS with PIL (out of scope);
what matters for the hot path is the *shape protocol* of ``Kinetics.__getitem__``
(kinetics_multigrid.py:214-259): the index carries (DataLoader task index, (sample, long-cycle
state)), the clip comes back as float32 [3, T, H, W] with
    (T, H=W) = f(long-cycle state, task index % 2 or % 3)
together with ``(clip, target, long_cycle_state, stats)``.  ``SyntheticKinetics`` keeps exactly
that protocol and fills the clip with N(0,1) noise (post-Normalize Kinetics frames are about
zero-mean / unit-variance); ``device_batch`` produces a whole step's batch directly in HBM.
"""
import torch

from cycle_batch_sampler import long_cycle_shapes, step_clip_shape


class SyntheticKinetics(torch.utils.data.Dataset):
    def __init__(self, n_samples=220000, n_classes=400, sample_duration=80, gamma_tau=5, crop_size=224, seed=0):
        self.n_samples, self.n_classes = n_samples, n_classes
        self.sample_duration, self.gamma_tau, self.crop_size = sample_duration, gamma_tau, crop_size
        self.long_cycles = long_cycle_shapes(sample_duration, crop_size)
        self.seed = seed

    def __len__(self):
        return self.n_samples

    def __getitem__(self, index):
        iteration = index[0]
        idx, long_cycle_state = index[1]
        frames, crop = self.long_cycles[long_cycle_state]
        stats = (frames, crop // 2, int(crop / 2 ** 0.5), crop)
        T, H = step_clip_shape(long_cycle_state, iteration, self.sample_duration, self.gamma_tau, self.crop_size)
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        clip = torch.randn(3, T, H, H, generator=g)
        target = int(torch.randint(0, self.n_classes, (1,), generator=g))
        return clip, target, long_cycle_state, stats


def device_batch(B, T, H, n_classes, device, generator=None):
    """One step's synthetic batch generated in HBM: clips float32[B,3,T,H,H], labels int64[B,1]."""
    x = torch.randn(B, 3, T, H, H, device=device, generator=generator)
    y = torch.randint(0, n_classes, (B, 1), device=device, generator=generator)
    return x, y


class DeviceVideoKinetics:
    """Kinetics.__getitem__'s protocol (kinetics_multigrid.py:214-259) over decoded uint8 videos that are already
    resident in HBM: the per-sample random draws happen on the host in the reference's order
    (x3dhip.clip_input.draw_clip_params), and crop + PIL-bilinear resize + flip + ToTensor + Normalize + the
    [3,T,H,W] stacking run as two HIP kernels for the whole batch (x3dhip.clip_input.ClipPreprocessor).

    videos: list of uint8 CUDA tensors [n_frames, H, W, 3]; labels: list of ints.  ``batch(indices, iteration,
    long_cycle_state)`` returns (clips float32 [B,3,T,S,S], labels int64 [B,1], long_cycle_state, stats)."""

    RESIZE = {'S': [180., 225.], 'M': [256., 320.], 'XL': [360., 450.]}      # train_x3d_kinetics_multigrid.py:54

    def __init__(self, videos, labels, sample_duration=80, gamma_tau=5, crop_size=224, x3d_version='M', rng=None):
        import random
        from x3dhip.clip_input import ClipPreprocessor
        self.videos, self.labels = videos, labels
        self.sample_duration, self.gamma_tau, self.crop_size = sample_duration, gamma_tau, crop_size
        self.long_cycles = long_cycle_shapes(sample_duration, crop_size)
        self.scales = [crop_size / i for i in self.RESIZE[x3d_version]]       # train...:70
        self.rng = rng if rng is not None else random
        self.pre = ClipPreprocessor(videos[0].device)

    def __len__(self):
        return len(self.videos)

    def batch(self, indices, iteration, long_cycle_state, out=None):
        from x3dhip.clip_input import draw_clip_params
        frames, crop = self.long_cycles[long_cycle_state]
        stats = (frames, crop // 2, int(crop / 2 ** 0.5), crop)
        T, S = step_clip_shape(long_cycle_state, iteration, self.sample_duration, self.gamma_tau, self.crop_size)
        samples = []
        for i in indices:
            v = self.videos[i]
            p = draw_clip_params(v.shape[0], v.shape[2], v.shape[1], self.scales, S, self.sample_duration, self.gamma_tau,
                                 frames, rng=self.rng)
            samples.append((v, p))
        clips = self.pre(samples, out=out)
        y = torch.tensor([[self.labels[i]] for i in indices], dtype=torch.int64, device=clips.device)
        return clips, y, long_cycle_state, stats

    def val_batch(self, indices, crops=3, sample_duration=None, crop_size=None):
        """The validation dataset's batch (kinetics.py:205-239 with the transforms of train...:132-136): for every video
        `crops` temporal windows, centre crop scaled to crop_size, no flip.  Returns (clips float32
        [B, crops, 3, T, S, S], labels int64 [B]) -- what train_x3d_kinetics_multigrid.validate consumes."""
        from x3dhip.clip_input import center_crop_box, val_crop_indices
        sd = self.sample_duration if sample_duration is None else sample_duration
        S = self.crop_size if crop_size is None else crop_size
        samples = []
        for i in indices:
            v = self.videos[i]
            x1, y1, crop = center_crop_box(v.shape[2], v.shape[1])
            for idx in val_crop_indices(v.shape[0], self.gamma_tau, sd, crops):
                samples.append((v, dict(frame_idx=idx, x1=x1, y1=y1, crop=crop, out=S, flip=False)))
        clips = self.pre(samples)
        B = len(indices)
        y = torch.tensor([self.labels[i] for i in indices], dtype=torch.int64, device=clips.device)
        return clips.view(B, crops, *clips.shape[1:]), y
