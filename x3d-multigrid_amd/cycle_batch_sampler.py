"""Multigrid long/short-cycle batch scheduler -- drop-in for the reference's
``cycle_batch_sampler.py`` (RandomEpochSampler :4-25, CycleBatchSampler :28-113).

Same constructor signatures, same attributes (``iteration_counter``, ``phase``,
``long_cycle_index`` ...) and the same stream of batches -- lists of
``(sample_index, long_cycle_index)`` whose length is the step's batch size -- as the reference
class, including its quirks (five fast-forward steps when an iterator starts, the ``>`` vs ``>=``
asymmetry at phase boundaries, IndexError when iterated past the last schedule entry).
The integer logic lives in ``MultigridSchedule`` (no torch), which the synthetic training driver
uses directly to obtain ``(B, T, H, W)`` per step without materialising index lists.
"""
import math

import torch
from torch.utils.data import sampler


class MultigridSchedule:
    """Step-wise state machine of the long cycle (which of the 4 shapes) and the short cycle
    (batch multiplier 2,1 or 4,2,1)."""

    def __init__(self, batch_size, schedule, cur_iterations, long_cycle_bs_scale):
        self.batch_size = batch_size
        self.schedule = schedule
        self.long_cycle_bs_scale = long_cycle_bs_scale
        self.iteration_counter = cur_iterations
        self.short_iteration_counter = 0
        self.phase = 1
        self.phase_steps = self._span(1)
        self.long_cycle_index = 0
        self.iter_offset = 0

    def _span(self, phase):
        return (self.schedule[phase] - self.schedule[phase - 1]) / len(self.long_cycle_bs_scale)

    def _last_phase(self):
        return self.phase == len(self.schedule) - 1

    def _scaled(self):
        return self.batch_size * self.long_cycle_bs_scale[self.long_cycle_index]

    def adjust_long_cycle(self, batch_size):
        """Advance phase / long-cycle index for the current iteration counter; returns the
        (possibly new) long-cycle batch size."""
        boundary = self.schedule[self.phase]          # IndexError past the end, as the reference
        if self.iteration_counter > boundary:
            self.iter_offset = boundary
            self.phase += 1
            self.phase_steps = self._span(self.phase)
            self.long_cycle_index = -1 if self._last_phase() else 0
            return self._scaled()
        if self.iteration_counter >= self.phase_steps + self.iter_offset:
            self.iter_offset += self.phase_steps
            nxt = -1 if self._last_phase() else self.long_cycle_index + 1
            self.long_cycle_index = min(nxt, len(self.long_cycle_bs_scale) - 1)
            return self._scaled()
        return batch_size

    def adjust_short_cycle(self, batch_size):
        if self.long_cycle_index in (0, 1):
            mult = (2, 1)[self.short_iteration_counter % 2]
        else:
            mult = (4, 2, 1)[self.short_iteration_counter % 3]
        return batch_size * mult

    def start(self):
        """State at the beginning of an iterator: returns (long batch size, first step batch)."""
        bs = self._scaled()
        self.short_iteration_counter = 0
        for _ in range(5):
            bs = self.adjust_long_cycle(bs)
        return bs, self.adjust_short_cycle(bs)

    def advance(self, bs):
        self.iteration_counter += 1
        self.short_iteration_counter += 1
        bs = self.adjust_long_cycle(bs)
        return bs, self.adjust_short_cycle(bs)

    def steps(self):
        """Generator of (step_batch_size, long_cycle_index, short_iteration_counter)."""
        bs, cur = self.start()
        while True:
            yield cur, self.long_cycle_index, self.short_iteration_counter
            bs, cur = self.advance(bs)


class RandomEpochSampler(sampler.RandomSampler):
    """Endless stream of fresh random permutations of the dataset (reference :4-25)."""

    def __init__(self, data_source, replacement=False, num_samples=None, epochs=1):
        self.epochs = epochs
        super().__init__(data_source, replacement, num_samples)

    @property
    def num_samples(self):
        base = len(self.data_source) if self._num_samples is None else self._num_samples
        return base * self.epochs

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        n = len(self.data_source)
        while True:
            yield from torch.randperm(n).tolist()


class CycleBatchSampler(sampler.BatchSampler):
    """BatchSampler whose batch length follows the multigrid schedule; every element is
    ``(sample_index, long_cycle_index)`` so the dataset can pick (T, crop) (reference :28-74)."""

    def __init__(self, sampler, batch_size, drop_last, schedule, cur_iterations, long_cycle_bs_scale):
        super().__init__(sampler, batch_size, drop_last)
        self._ms = MultigridSchedule(batch_size, schedule, cur_iterations, long_cycle_bs_scale)
        self.schedule = schedule
        self.long_cycle_bs_scale = long_cycle_bs_scale

    # the reference exposes these as plain attributes; keep them readable/writable
    def _fwd(name):
        return property(lambda self: getattr(self._ms, name), lambda self, v: setattr(self._ms, name, v))

    iteration_counter = _fwd("iteration_counter")
    short_iteration_counter = _fwd("short_iteration_counter")
    phase = _fwd("phase")
    phase_steps = _fwd("phase_steps")
    long_cycle_index = _fwd("long_cycle_index")
    iter_offset = _fwd("iter_offset")
    del _fwd

    def adjust_long_cycle(self, batch_size):
        return self._ms.adjust_long_cycle(batch_size)

    def adjust_short_cycle(self, batch_size):
        return self._ms.adjust_short_cycle(batch_size)

    def __iter__(self):
        bs, want = self._ms.start()
        batch = []
        for idx in self.sampler:
            batch.append((idx, self._ms.long_cycle_index))
            if len(batch) == want:
                yield batch
                batch = []
                bs, want = self._ms.advance(bs)
        if batch and not self.drop_last:
            yield batch


def long_cycle_shapes(sample_duration, crop_size=224):
    """(frames, crop) per long-cycle index (kinetics_multigrid.py:205-209)."""
    small = int(math.floor(crop_size / math.sqrt(2)))
    return [(sample_duration // 4, small), (sample_duration // 2, small),
            (sample_duration // 2, crop_size), (sample_duration, crop_size)]


def step_clip_shape(long_cycle_index, task_index, sample_duration, gamma_tau, crop_size=224):
    """(T, H=W) of the clip fed to the model at DataLoader task ``task_index``
    (kinetics_multigrid.py:225-237 + temporal_transforms.py:101-110)."""
    frames, crop = long_cycle_shapes(sample_duration, crop_size)[long_cycle_index]
    if long_cycle_index in (0, 1):
        if task_index % 2 == 0:
            crop = int(math.floor(crop / math.sqrt(2)))
    else:
        phase = task_index % 3
        if phase == 0:
            crop = crop // 2
        elif phase == 1:
            crop = int(math.floor(crop / math.sqrt(2)))
    return frames // gamma_tau, crop
