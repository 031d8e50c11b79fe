"""Charades localisation losses of the reference's `train_x3d_charades_loc.py` (:123, :168-189) for the per-frame head
(`generate_model(..., task='loc')`, x3d.py:340-343), computed by one HIP kernel pair (csrc/head.hip: x3d_loc_losses):

    per_frame_logits = F.interpolate(x3d(inputs), tl, mode='linear')              # [B, C, TL]
    cls_loss = BCEWithLogits(per_frame_logits.max(2)[0], labels.max(2)[0])
    loc_loss = BCEWithLogits(per_frame_logits, labels)
    loss     = (cls_loss + loc_loss) / (2 * num_steps_per_update)

The training scripts, datasets, mAP meters and annotations of the Charades pipeline are out of scope (SURVEY.md section 2
rows 6-13); this is the arithmetic on the model's output that the hot path ends in.
"""
import torch

from x3dhip import ops


class _LocLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, num_steps_per_update):
        losses, dlog = ops.loc_losses(logits.contiguous().float(), labels.contiguous().float(),
                                      grad_scale=1.0 / (2.0 * num_steps_per_update))
        ctx.save_for_backward(dlog)
        ctx.k = num_steps_per_update
        ctx.mark_non_differentiable(losses)
        return (losses[0] + losses[1]) / (2.0 * num_steps_per_update), losses

    @staticmethod
    def backward(ctx, gloss, _glosses):
        (dlog,) = ctx.saved_tensors
        return dlog * gloss, None, None


def charades_loc_loss(per_frame_logits, labels, num_steps_per_update=1):
    """per_frame_logits [B, C, T] (the model's output, NOT yet interpolated), labels float [B, C, TL].
    Returns (loss, cls_loss, loc_loss); loss is differentiable w.r.t. the logits."""
    loss, losses = _LocLossFunction.apply(per_frame_logits, labels, num_steps_per_update)
    return loss, losses[0], losses[1]
