"""Multigrid training entry point for X3D on MI355X -- mirror of the reference's
``train_x3d_kinetics_multigrid.py`` for the hot path (constants :49-61, setup_data :64-105,
run :108-296, lr_warmup :300-305, print_stats :308-315).

What is the same: the module constants and their meaning, the long-cycle switch (BN splits
re-created, LR scaled by LONG_CYCLE on (re)start / last cycle and LONG_CYCLE_LR_SCALE otherwise),
warm-up towards the *scaled* LR, MultiStepLR milestones with the second-to-last moved to the
middle of the last phase, SGD(momentum .9, wd 5e-5 on every parameter), CE on logits[B,C,1],
checkpoints {'model_state_dict','optimizer_state_dict','scheduler_state_dict','long_ind'}
every 4000 steps with split-BN re-shaping before load.

What differs (MI355X-first): one process per GPU (``torch.distributed`` over RCCL) instead of
nn.DataParallel -- BATCH is split evenly over ranks, gradients are all-reduced as one flat
buffer and averaged in the fused SGD kernel; BN statistics stay rank-local exactly as
DataParallel replicas' do.  Clips are synthetic NCTHW tensors generated in HBM
(``--synthetic``, the only mode: JPEG loading is out of scope); the per-step (B,T,H,W) comes
from the same schedule arithmetic as the reference's sampler + dataset.

    python train_x3d_kinetics_multigrid.py -gpu 0 --steps 60 --iters-per-epoch 40 --max-epochs 3
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        train_x3d_kinetics_multigrid.py --steps 200
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x3d as resnet_x3d  # noqa: E402
import cycle_batch_sampler as cbs  # noqa: E402
from kinetics_multigrid import device_batch  # noqa: E402

KINETICS_DATASET_SIZE = {'train': 220000, 'val': 17500}

BS = 8
BS_UPSCALE = 16  # CHANGE WITH GPU AVAILABILITY
INIT_LR = (1.6 / 1024) * (BS * BS_UPSCALE)
SCHEDULE_SCALE = 4
EPOCHS = (60000 * 1024 * 1.5) / 220000

LONG_CYCLE = [8, 4, 2, 1]
LONG_CYCLE_LR_SCALE = [8, 0.5, 0.5, 0.5]
GPUS = 4
BASE_BS_PER_GPU = BS * BS_UPSCALE // GPUS  # FOR SPLIT BN
CONST_BN_SIZE = 8

X3D_VERSION = 'M'  # ['S', 'M', 'XL']


def lr_schedule_milestones(num_iterations):
    """schedule[1:] with the second-to-last entry moved to the midpoint of the last phase."""
    schedule = [int(f * num_iterations) for f in (0, 0.4, 0.65, 0.85, 1)]
    milestones = list(schedule)
    milestones[-2] = (milestones[-2] + milestones[-1]) // 2
    return schedule, milestones[1:]


def setup_data(batch_size, num_steps_per_update, epochs, iterations_per_epoch, cur_iterations, crop_size,
               resize_size, num_frames, gamma_tau):
    """Returns (step-shape generator, None, lr milestones).  The generator yields
    (global_batch, long_ind, (T, H)) per optimizer step -- what the reference's
    CycleBatchSampler + DataLoader + Kinetics.__getitem__ chain produces."""
    num_iterations = int(epochs * iterations_per_epoch)
    schedule, milestones = lr_schedule_milestones(num_iterations)
    ms = cbs.MultigridSchedule(batch_size, schedule, cur_iterations, LONG_CYCLE)

    def shapes():
        for n, long_ind, task in ms.steps():
            yield n, long_ind, cbs.step_clip_shape(long_ind, task, num_frames, gamma_tau)

    return shapes(), None, milestones


def lr_warmup(init_lr, cur_steps, warmup_steps, opt):
    start_after = 1
    if start_after < cur_steps < warmup_steps:
        lr_scale = min(1., float(cur_steps + 1) / warmup_steps)
        for pg in opt.param_groups:
            pg['lr'] = lr_scale * init_lr


class MultiStepLR:
    """Chainable MultiStepLR (torch.optim.lr_scheduler.MultiStepLR semantics, train...:184):
    each ``step()`` multiplies the current lr by gamma when the new epoch is a milestone."""

    def __init__(self, opt, milestones, gamma=0.1, last_epoch=0):
        self.opt, self.milestones, self.gamma, self.last_epoch = opt, list(milestones), gamma, last_epoch

    def step(self):
        self.last_epoch += 1
        k = self.milestones.count(self.last_epoch)
        if k:
            for pg in self.opt.param_groups:
                pg['lr'] *= self.gamma ** k

    def state_dict(self):
        return {'milestones': {m: self.milestones.count(m) for m in self.milestones}, 'gamma': self.gamma,
                'last_epoch': self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd['last_epoch']
        self.gamma = sd.get('gamma', self.gamma)


def print_stats(long_ind, batch_size, stats, gamma_tau, bn_splits, lr):
    bs = batch_size * LONG_CYCLE[long_ind]
    if long_ind in [0, 1]:
        print(' ***** LR {} Frames {}/{} BS ({},{}) W/H ({},{}) BN_splits {} long_ind {} *****'.format(
            lr, stats[0], gamma_tau, bs * 2, bs, stats[2], stats[3], bn_splits, long_ind), flush=True)
    else:
        print(' ***** LR {} Frames {}/{} BS ({},{},{}) W/H ({},{},{}) BN_splits {} long_ind {} *****'.format(
            lr, stats[0], gamma_tau, bs * 4, bs * 2, bs, stats[1], stats[2], stats[3], bn_splits, long_ind), flush=True)


def validate(model, batches):
    """Validation phase of the reference loop (train_x3d_kinetics_multigrid.py:203-206, 239-266, 288-292):
    eval mode, `aggregate_sub_bn_stats()` first, no gradients; every batch is [b, n, 3, T, H, W] with n temporal crops
    per video, run as b*n clips; prediction = argmax of the crop-averaged SOFTMAX, loss = cross entropy of the
    crop-averaged LOGITS against the label.  Returns (mean loss per batch, top-1 accuracy, videos seen).
    The model is left in eval mode (the caller switches back with model.train(True), as the reference does)."""
    import torch.nn.functional as F
    model.train(False)
    model.aggregate_sub_bn_stats()
    tot_cls_loss, tot_corr, tot_dat, num_iter = 0.0, 0.0, 0, 0
    with torch.no_grad():
        for inputs, labels in batches:
            num_iter += 1
            b, n, c, t, h, w = inputs.shape
            logits = model(inputs.view(b * n, c, t, h, w))                 # [b*n, classes, 1]
            logits = logits.view(b, n, logits.shape[1], 1)
            logits_sm = torch.mean(F.softmax(logits, dim=2), 1)
            logits = torch.mean(logits, 1)
            preds = torch.max(logits_sm, 1)[1]                             # [b, 1]
            labels = labels.view(b, 1)
            tot_cls_loss += float(F.cross_entropy(logits, labels))
            tot_corr += float(torch.sum(preds == labels))
            tot_dat += b
    return tot_cls_loss / max(num_iter, 1), tot_corr / max(tot_dat, 1), tot_dat


def _save_ckpt(model, optimizer, lr_sched, long_ind, save_model, steps):
    """The reference's checkpoint record (train_x3d_kinetics_multigrid.py:286-291)."""
    ckpt = {'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(),
            'scheduler_state_dict': lr_sched.state_dict(), 'long_ind': long_ind}
    os.makedirs(os.path.dirname(save_model) or '.', exist_ok=True)
    torch.save(ckpt, save_model + str(steps).zfill(6) + '.pt')


def run(init_lr=INIT_LR, warmup_steps=8000, max_epochs=120, batch_size=BS * BS_UPSCALE, steps=0, max_steps_run=None,
        iterations_per_epoch=None, load_ckpt=None, save_model='models/x3d_multigrid_kinetics_rgb_sgd_',
        save_every=4000, use_graph=True, x3d_version=X3D_VERSION, log_every=20, val_every=None, val_batches=2,
        val_batch_size=2, val_crops=3, num_steps_per_update=1, clip_size=None, act_dtype=torch.float32):
    """The reference's training loop (train_x3d_kinetics_multigrid.py:157-292) on synthetic clips.  val_every: run the
    validation phase (`validate`, the reference does it after every 4 training epochs, :195) every that many steps on
    `val_batches` synthetic batches of [val_batch_size, val_crops, 3, T, H, W].
    num_steps_per_update: gradient accumulation over that many micro-batches per optimizer step (train...:119,267-273;
    the schedule then counts iterations and lr_schedule is divided by it, :130).  clip_size overrides the crop size of
    the shape table (tests: the default batch arithmetic at a tiny resolution)."""
    from x3dhip.trainer import Trainer
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD

    frames = 80
    crop_size = {'S': 160, 'M': 224, 'XL': 312, 'L': 312}[x3d_version]
    resize_size = {'S': [180., 225.], 'M': [256., 256.], 'XL': [360., 450.], 'L': [360., 450.]}[x3d_version]
    gamma_tau = {'S': 6, 'M': 5, 'XL': 5, 'L': 5}[x3d_version]
    st_steps = load_steps = steps
    if batch_size % world != 0:
        raise ValueError("global batch %d is not divisible by the world size %d" % (batch_size, world))
    if iterations_per_epoch is None:
        iterations_per_epoch = KINETICS_DATASET_SIZE['train'] // batch_size
    last_long = -2

    shapes, _, lr_schedule = setup_data(batch_size, num_steps_per_update, max_epochs, iterations_per_epoch,
                                        steps * num_steps_per_update, crop_size, resize_size, frames, gamma_tau)
    lr_schedule = [i // num_steps_per_update for i in lr_schedule]          # train...:130
    total_steps = lr_schedule[-1]
    if rank == 0:
        print('Total iterations:', lr_schedule[-1] * num_steps_per_update, 'Total steps:', lr_schedule[-1])

    base_per_gpu = batch_size // world
    base_splits = max(1, base_per_gpu // CONST_BN_SIZE)
    # act_dtype=torch.bfloat16: mixed-storage mode (bf16 storage of the wide bottleneck tensors, fp32 arithmetic; not in the
    # reference, BASELINE config 5)
    model = resnet_x3d.generate_model(x3d_version=x3d_version, n_classes=400, n_input_channels=3, dropout=0.5,
                                      base_bn_splits=base_splits, act_dtype=act_dtype)
    ck = None
    if load_ckpt is not None:
        ck = torch.load(load_ckpt, map_location='cpu')
        cur_long_ind = ck['long_ind']
        model.update_bn_splits_long_cycle(LONG_CYCLE[cur_long_ind])   # split_bn buffers are [C*S]
        model.load_state_dict(ck['model_state_dict'])
        last_long = cur_long_ind
    model.to(dev).train(True)

    lr = init_lr
    optimizer = Trainer(model, lr=lr, momentum=0.9, weight_decay=5e-5, process_group=pg, world_size=world,
                        use_graph=use_graph, num_steps_per_update=num_steps_per_update)
    lr_sched = MultiStepLR(optimizer, lr_schedule)
    if ck is not None:
        optimizer.load_state_dict(ck['optimizer_state_dict'])
        lr_sched.load_state_dict(ck['scheduler_state_dict'])
        lr = optimizer.param_groups[0]['lr']

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    tot_loss = tot_corr = tot_dat = 0.0
    t0 = time.time()
    clips = 0
    done = 0
    bn_splits = model.bn1.num_splits
    long_ind = max(last_long, 0)
    try:
        for n_global, long_ind, (T, H) in shapes:
            # the reference's loop ends with max_epochs (train...:192); the schedule generator itself has no end, and
            # its long-cycle lookup runs off the table one step past schedule[-1]
            if (max_steps_run is not None and done >= max_steps_run) or steps >= total_steps:
                break
            if long_ind != last_long:
                bn_splits = model.update_bn_splits_long_cycle(LONG_CYCLE[long_ind])
                optimizer.invalidate_graphs()   # captured graphs point at the split_bn buffers that were just re-created
                lr_scale_fact = LONG_CYCLE[long_ind] if (last_long == -2 or long_ind == -1) else LONG_CYCLE_LR_SCALE[long_ind]
                last_long = long_ind
                for g in optimizer.param_groups:
                    g['lr'] *= lr_scale_fact
                    lr = g['lr']
                if rank == 0:
                    fr, cr = cbs.long_cycle_shapes(frames, 224)[long_ind]
                    print_stats(long_ind, batch_size, (fr // gamma_tau, cr // 2, int(cr / 2 ** 0.5), cr), gamma_tau,
                                bn_splits, lr)
            if n_global % world != 0:
                raise ValueError("step batch %d is not divisible by the world size %d" % (n_global, world))
            B = n_global // world
            if clip_size is not None:
                H = max(8, H * clip_size // {'S': 160, 'M': 224, 'XL': 312, 'L': 312}[x3d_version])
            inputs, labels = device_batch(B, T, H, 400, dev, gen)
            loss, logits = optimizer.train_step(
                inputs, labels, pre_step=lambda: lr_warmup(lr, steps - st_steps, warmup_steps, optimizer))
            clips += n_global
            if not optimizer.stepped:           # a micro-batch of an accumulated step (num_steps_per_update > 1)
                continue
            steps += 1
            done += 1
            lr_sched.step()
            if done % log_every == 0 or done == 1:
                tot_loss = float(loss)
                preds = logits.argmax(1)
                acc = float((preds == labels).float().mean())
                if rank == 0:
                    dt = time.time() - t0
                    print(' step {} long {} shape ({},{},{}) loss {:.4f} acc {:.3f} lr {:.5f}  {:.1f} clips/s'.format(
                        steps, long_ind, B, T, H, tot_loss, acc, optimizer.param_groups[0]['lr'], clips / dt), flush=True)
            if val_every and done % val_every == 0:
                Tv, Hv = frames // gamma_tau, crop_size
                vb = []
                for _ in range(val_batches):
                    xv, yv = device_batch(val_batch_size * val_crops, Tv, Hv, 400, dev, gen)
                    vb.append((xv.view(val_batch_size, val_crops, 3, Tv, Hv, Hv), yv.view(-1)[:val_batch_size]))
                v_loss, v_acc, v_seen = validate(model, vb)
                model.train(True)                                   # train...:199-200
                if rank == 0:
                    print(' val after step {}: Cls Loss: {:.4f} Acc: {:.4f} ({} videos)'.format(steps, v_loss, v_acc, v_seen),
                          flush=True)
            if save_every and steps % save_every == 0 and rank == 0:
                _save_ckpt(model, optimizer, lr_sched, long_ind, save_model, steps)
        if save_every and rank == 0 and steps >= total_steps and done > 0 and steps % save_every != 0:
            _save_ckpt(model, optimizer, lr_sched, long_ind, save_model, steps)     # end of the schedule: final checkpoint
    finally:
        # runs on errors too: a rank that raised must not leave the others waiting inside a collective
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
    return steps, clips / max(time.time() - t0, 1e-9)


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-gpu', default=None, type=str, help='CUDA_VISIBLE_DEVICES (single-process runs)')
    parser.add_argument('--synthetic', action='store_true', default=True)
    parser.add_argument('--steps', type=int, default=100, help='optimizer steps to run in this invocation')
    parser.add_argument('--start-step', type=int, default=0)
    parser.add_argument('--batch', type=int, default=BS * BS_UPSCALE, help='GLOBAL base batch (reference: 128)')
    parser.add_argument('--max-epochs', type=int, default=120)
    parser.add_argument('--iters-per-epoch', type=int, default=None)
    parser.add_argument('--warmup-steps', type=int, default=8000)
    parser.add_argument('--load', default=None)
    parser.add_argument('--save-every', type=int, default=4000)
    parser.add_argument('--no-graph', action='store_true')
    parser.add_argument('--version', default=X3D_VERSION)
    parser.add_argument('--bf16', action='store_true', help='bf16 storage of the wide bottleneck tensors (fp32 arithmetic)')
    args = parser.parse_args()
    if args.gpu is not None:
        os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu
    run(init_lr=(1.6 / 1024) * args.batch, warmup_steps=args.warmup_steps, max_epochs=args.max_epochs,
        batch_size=args.batch, steps=args.start_step, max_steps_run=args.steps,
        iterations_per_epoch=args.iters_per_epoch, load_ckpt=args.load, save_every=args.save_every,
        use_graph=not args.no_graph, x3d_version=args.version,
        act_dtype=torch.bfloat16 if args.bf16 else torch.float32)
