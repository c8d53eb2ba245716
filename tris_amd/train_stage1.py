"""Stage-1 training step on MI355X -- the hot loop of the reference's train_stage1.py:286-411 (`train_one_epoch`),
plus `clip_forward` / `MaxLoss` (:263-284) under the same names.

`train_step` is one iteration (forward TRIS -> CLIP-guided fg / negative-sample / cls losses -> backward ->
[gradient all-reduce] -> AdamW -> LR schedule).  Numerically it is the reference's step; structurally it is the
"lean" form: the auxiliary ViT-B/32 runs ONCE on the foreground image (the reference runs it twice on
the same input, :340 and :344), all positive + negative sentences go through the frozen aux text encoder in ONE batch
(the reference loops over images, :346-347), no weight gradients are computed for the frozen aux CLIP, and the unused
attention pool of the RN50 is skipped.  Loss values and gradients are identical (tests/test_gpu_parity.py).
"""
import time

import torch

from . import ops
from .loss.clip_loss import clip_forward  # noqa: F401  (same import surface as the reference module)

CLIP_INPUT = 224


def MaxLoss(x):
    """-mean(log(clamp(x, 1e-4, 0.9999)))  (train_stage1.py:280-284); small host-side helper for API parity"""
    return -(torch.log(x.clamp(0.0001, 0.9999))).mean()


def stage1_forward_losses(model, clip_model, img, word_ids, neg_word_ids, args):
    """train_stage1.py:317-364.  Returns (losses[4] = total,l1,l4,l5 ; cls ; sig_out)."""
    B = img.shape[0]
    # frozen aux text tower (positives + negatives in one batch, no gradient): issued first, on the side stream, so it
    # overlaps the RN50 trunk of the model forward
    from .model.model_stage1 import _overlap_enabled, _side_stream
    f_all = None
    ids_all = word_ids.long()
    K = 0
    if neg_word_ids is not None and args.negative_samples > 0:
        K = neg_word_ids.shape[1]
        ids_all = torch.cat([ids_all, neg_word_ids.long().reshape(B * K, -1)], 0)
    if _overlap_enabled():
        main, side = torch.cuda.current_stream(), _side_stream(img.device)
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            f_all = clip_model.encode_text(ids_all)[1]
    cls, _, _, sig_out, _ = model(img, word_ids)
    if img.shape[2] != CLIP_INPUT:
        cam = ops.resize_bilinear(sig_out, (CLIP_INPUT, CLIP_INPUT), True)
        with torch.no_grad():
            im = ops.resize_bilinear(img, (CLIP_INPUT, CLIP_INPUT), True)
    else:
        cam, im = sig_out, img
    vit = clip_model.visual
    f_i = vit.forward_patches(ops.fg_patches(cam, im, vit.patch_size))
    with torch.no_grad():
        if f_all is None:
            f_all = clip_model.encode_text(ids_all)[1]
        else:
            torch.cuda.current_stream().wait_stream(_side_stream(img.device))
            f_all.record_stream(torch.cuda.current_stream())
        f_t = f_all[:B].contiguous()
        f_neg = f_all[B:].reshape(B, K, -1).contiguous() if K > 0 else None
    losses = ops.stage1_loss(cls, f_i, f_t, f_neg, float(args.w1), float(args.w4), float(args.w5))
    return losses, cls, sig_out


def train_step(model, clip_model, optimizer, img, word_ids, neg_word_ids, args, lr_scheduler=None, reducer=None):
    """One optimisation step; returns the device tensor losses[4] (no host sync)."""
    losses, _, _ = stage1_forward_losses(model, clip_model, img, word_ids, neg_word_ids, args)
    optimizer.zero_grad()
    losses[0].backward()
    if reducer is not None:
        reducer.reduce()
    optimizer.step()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return losses


def freeze_aux(clip_model):
    """The aux CLIP is never optimised (train_stage1.py:167-168); drop its (wasted) weight gradients."""
    clip_model.eval()
    for p in clip_model.parameters():
        p.requires_grad_(False)
    return clip_model


def train_one_epoch(train_loader, model, optimizer, epoch, local_rank, args, iteration=0, clip_model=None,
                    lr_scheduler=None, reducer=None, logger=None, writer=None):
    """Same signature and batch-dict contract as the reference (train_stage1.py:286); returns `iteration`."""
    model.train()
    freeze_aux(clip_model)
    num_steps = len(train_loader)
    t0 = time.time()
    last = None
    for idx, (samples, targets) in enumerate(train_loader):
        word_ids = samples["word_ids"].squeeze(1).cuda(local_rank, non_blocking=True)
        img = samples["img"].cuda(local_rank, non_blocking=True)
        neg = samples["neg_word_ids"].cuda(local_rank, non_blocking=True) if args.negative_samples > 0 else None
        last = train_step(model, clip_model, optimizer, img, word_ids, neg, args, lr_scheduler, reducer)
        if idx % args.print_freq == 0 and local_rank == 0:
            v = last.tolist()  # the only host sync, every print_freq steps (the reference syncs every step, :374-387)
            msg = (f"Train:[{epoch:2d}/{args.epoch}][{idx:4d}/{num_steps}] | lr {optimizer.param_groups[0]['lr']:.6f} || "
                   f"loss: {v[0]:.4f} | l1: {v[1]:.4f} | l4: {v[2]:.4f} | l5: {v[3]:.4f} | "
                   f"time/step: {(time.time() - t0) / (idx + 1):.4f}")
            (logger.info if logger is not None else print)(msg)
            if writer is not None:
                for name, val in zip(("train/loss", "train/l1", "train/l4", "train/l5"), v):
                    writer.add_scalar(name, val, iteration)
        iteration += 1
    return iteration
