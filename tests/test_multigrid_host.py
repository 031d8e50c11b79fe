"""Host-side multigrid logic of the product (no GPU): the drop-in CycleBatchSampler against the
golden sequences recorded from the reference's class, the shape table, and the training script's
LR rules against the oracle's restatement."""
import os

import numpy as np
import pytest
import torch

import cycle_batch_sampler as cbs
from oracle import multigrid_oracle as mo


class _Endless:
    def __iter__(self):
        i = 0
        while True:
            yield i
            i += 1

    def __len__(self):
        return 1 << 30


def _run(batch, schedule, cur, n):
    s = cbs.CycleBatchSampler(_Endless(), batch, False, schedule=list(schedule), cur_iterations=cur,
                              long_cycle_bs_scale=[8, 4, 2, 1])
    it = iter(s)
    lens, longs = [], []
    for _ in range(n):
        b = next(it)
        lens.append(len(b))
        longs.append(b[0][1])
        assert all(e[1] == b[0][1] for e in b)
    return lens, longs, s


def test_cycle_batch_sampler_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    sch = list(g["schedule_full"])
    lens, longs, _ = _run(128, sch, 0, 12)
    assert lens == list(g["full_first_len"]) and longs == list(g["full_first_long"])
    lens, longs, _ = _run(128, sch, 204000, 12)
    assert lens == list(g["full_resume_len"]) and longs == list(g["full_resume_long"])
    small = list(g["schedule_small"])
    lens, longs, s = _run(8, small, 0, 400)
    assert lens == list(g["small_len"]) and longs == list(g["small_long"])
    assert s.iteration_counter == 399 + 0 and s.phase == 4     # attributes readable like the reference's
    lens, longs, _ = _run(8, small, 200, 150)
    assert lens == list(g["small_resume_len"]) and longs == list(g["small_resume_long"])
    # transition table of the full schedule via the torch-free state machine
    ms = cbs.MultigridSchedule(1, sch, 0, [8, 4, 2, 1])
    st = ms.steps()
    longs = np.array([next(st)[1] for _ in range(206100)])
    chg = np.nonzero(np.diff(longs))[0] + 1
    assert list(chg) == list(g["full_transitions_at"]) and list(longs[chg]) == list(g["full_transitions_to"])
    with pytest.raises(IndexError):
        ms2 = cbs.MultigridSchedule(1, sch, 206100, [8, 4, 2, 1])
        st2 = ms2.steps()
        for _ in range(200):
            next(st2)


def test_drop_last_and_short_tail():
    s = cbs.CycleBatchSampler(range(20), 1, False, schedule=[0, 160, 260, 340, 400], cur_iterations=0,
                              long_cycle_bs_scale=[8, 4, 2, 1])
    got = [len(b) for b in s]
    assert got == [16, 4]          # 16 = 1*8*2, then the 4 left-overs
    s = cbs.CycleBatchSampler(range(20), 1, True, schedule=[0, 160, 260, 340, 400], cur_iterations=0,
                              long_cycle_bs_scale=[8, 4, 2, 1])
    assert [len(b) for b in s] == [16]


def test_random_epoch_sampler_is_endless_permutations():
    r = cbs.RandomEpochSampler(range(5), epochs=3)
    assert len(r) == 15
    it = iter(r)
    a = [next(it) for _ in range(10)]
    assert sorted(a[:5]) == list(range(5)) and sorted(a[5:]) == list(range(5))


def test_shape_table_matches_oracle_and_survey():
    for li in (0, 1, 2, 3, -1):
        for task in range(6):
            for frames, gt, crop in ((80, 5, 224), (80, 10, 224), (80, 6, 224)):
                assert cbs.step_clip_shape(li, task, frames, gt, crop) == mo.step_shape(li, task, frames, gt, crop)
    assert cbs.long_cycle_shapes(80) == [(20, 158), (40, 158), (40, 224), (80, 224)]
    assert [cbs.step_clip_shape(3, t, 80, 5) for t in range(3)] == [(16, 112), (16, 158), (16, 224)]


def test_training_script_lr_rules():
    import train_x3d_kinetics_multigrid as tr
    assert tr.BASE_BS_PER_GPU // tr.CONST_BN_SIZE == 4 and tr.INIT_LR == pytest.approx(0.2)
    sch, ms = tr.lr_schedule_milestones(206160)
    assert sch == [0, 82464, 134004, 175236, 206160] and ms == mo.lr_milestones(206160)

    class Opt:
        param_groups = [{'lr': 1.0}]
    o = Opt()
    for cur in (0, 1, 2, 100, 7998, 7999, 8000, 9000):
        o.param_groups[0]['lr'] = 123.0
        tr.lr_warmup(1.6, cur, 8000, o)
        want = mo.warmup_lr(1.6, cur, 8000)
        assert o.param_groups[0]['lr'] == (123.0 if want is None else pytest.approx(want))
    # chainable MultiStepLR vs torch's
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=0.8)
    tsch = torch.optim.lr_scheduler.MultiStepLR(topt, [3, 5, 5, 9])
    o.param_groups[0]['lr'] = 0.8
    mine = tr.MultiStepLR(o, [3, 5, 5, 9])
    for _ in range(12):
        topt.step(); tsch.step(); mine.step()
        assert o.param_groups[0]['lr'] == pytest.approx(topt.param_groups[0]['lr'])
    # long-cycle LR factors (train...:229)
    for last, li in ((-2, 0), (-2, 2), (0, 1), (1, 2), (2, 3), (3, 0), (3, -1)):
        f = tr.LONG_CYCLE[li] if (last == -2 or li == -1) else tr.LONG_CYCLE_LR_SCALE[li]
        assert f == mo.long_cycle_lr_factor(li, last)


def test_synthetic_dataset_protocol():
    from kinetics_multigrid import SyntheticKinetics
    ds = SyntheticKinetics(n_samples=10, sample_duration=80, gamma_tau=5)
    clip, target, state, stats = ds[(1, (3, 2))]        # task 1 of long cycle 2
    assert clip.shape == (3, 8, 158, 158) and state == 2 and stats == (40, 112, 158, 224)
    clip, _, _, _ = ds[(0, (3, 0))]
    assert clip.shape == (3, 4, 111, 111)
