"""Stage-1 training step on MI355X -- the hot loop of the reference's train_stage1.py:286-411 (`train_one_epoch`),
plus `clip_forward` / `MaxLoss` (:263-284) under the same names.

`train_step` is one iteration (forward TRIS -> CLIP-guided fg / negative-sample / cls losses -> backward ->
[gradient all-reduce] -> AdamW -> LR schedule).  Numerically it is the reference's step; structurally it is the
"lean" form: the auxiliary ViT-B/32 runs ONCE on the foreground image (the reference runs it twice on
the same input, :340 and :344), all positive + negative sentences go through the frozen aux text encoder in ONE batch
(the reference loops over images, :346-347), no weight gradients are computed for the frozen aux CLIP, and the unused
attention pool of the RN50 is skipped.  Loss values and gradients are identical (tests/test_gpu_parity.py).
"""
import datetime
import logging
import os
import random
import time

import numpy as np
import torch

from . import ops
from .config import cfg
from .loss.clip_loss import clip_forward  # noqa: F401  (same import surface as the reference module)

CLIP_INPUT = 224


def MaxLoss(x):
    """-mean(log(clamp(x, 1e-4, 0.9999)))  (train_stage1.py:280-284); small host-side helper for API parity"""
    return -(torch.log(x.clamp(0.0001, 0.9999))).mean()


def stage1_forward_losses(model, clip_model, img, word_ids, neg_word_ids, args):
    """train_stage1.py:317-364.  Returns (losses[4] = total,l1,l4,l5 ; cls ; sig_out)."""
    B = img.shape[0]
    # frozen aux text tower (positives + negatives in one batch, no gradient): on the side stream, see below
    from .graphs import frozen_text
    from .model.model_stage1 import _overlap_enabled, _side_stream
    f_all = None
    ids_all = word_ids.long()
    K = 0
    if neg_word_ids is not None and args.negative_samples > 0:
        K = neg_word_ids.shape[1]
        ids_all = ops.concat_i64(ids_all, neg_word_ids.long().reshape(B * K, -1))
    main = side = ready = None
    if _overlap_enabled():
        main, side = torch.cuda.current_stream(), _side_stream(img.device)
        ready = torch.cuda.Event()
        ready.record(main)           # ids_all is complete here
    cls, _, _, sig_out, _ = model(img, word_ids)
    if side is not None:
        # The frozen aux text tower has no backward and its output is needed only by the loss: it is ISSUED here, behind the
        # TRIS forward (whose own text encoder is needed sooner and shares the side stream), so that the host starts the
        # trunk at once instead of issuing ~130 small launches first; it executes under the rest of the forward.
        # Replayed from a hipGraph (tris_amd.graphs.frozen_text): one launch instead of ~130.
        side.wait_event(ready)
        with torch.cuda.stream(side), torch.no_grad():
            f_all = frozen_text(clip_model, ids_all)
    def joined():       # called where the loss needs the features: the aux ViT forward is issued (and runs) before the wait
        torch.cuda.current_stream().wait_stream(_side_stream(img.device))
        f_all.record_stream(torch.cuda.current_stream())
        return f_all
    losses = stage1_loss_block(clip_model, img, cls, sig_out, joined if f_all is not None else None, ids_all, K, args)
    return losses, cls, sig_out


def stage1_loss_block(clip_model, img, cls, sig_out, f_all, ids_all, K, args):
    """train_stage1.py:336-364 after the TRIS forward: foreground image -> aux ViT -> the three losses.  f_all: the frozen aux
    text features of all B + B*K sentences if the caller already has them (or a callable returning them: a side-stream join), else
    None."""
    B = img.shape[0]
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
            f_all = clip_model.encode_text_hidden(ids_all)
        elif callable(f_all):
            f_all = f_all()
        f_t = f_all[:B].contiguous()
        f_neg = f_all[B:].reshape(B, K, -1).contiguous() if K > 0 else None
    return ops.stage1_loss(cls, f_i, f_t, f_neg, float(args.w1), float(args.w4), float(args.w5))


def _step_body(model, clip_model, optimizer, img, word_ids, neg_word_ids, args, reducer, device_hyper=False,
               optimizer_step=True):
    """forward -> losses -> backward -> [all-reduce] -> AdamW: the part of a step that is kernel launches only (what
    tris_amd.graphs.GraphedTrainStep captures).  device_hyper: the optimiser reads lr / bias corrections from device memory."""
    ops.h2_begin_step()   # (h2 arithmetic: a fresh amax pool + the weights' amaxes; no-op otherwise)
    losses, _, _ = stage1_forward_losses(model, clip_model, img, word_ids, neg_word_ids, args)
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin_step()
    losses[0].backward()
    if reducer is not None:
        reducer.reduce()
    if optimizer_step:
        optimizer.step(device_hyper=device_hyper)
    else:
        ops.wgrad_join()
    ops.h2_end_step()
    # detached: a caller that keeps the returned tensor must not keep the step's autograd nodes alive with it (AccumulateGrad
    # nodes remember the stream they were created on: a stale one breaks a later stream capture of the step)
    return losses.detach()


def _graphable(model, optimizer, img, reducer):
    """may this step be replayed from a hipGraph?  Opt-in (TRIS_STEP_GRAPH=1): see train_step"""
    if cfg.step_graph not in ("1", "seg") or ops._PROF is not None:
        return False
    if not img.is_cuda or not hasattr(optimizer, "enable_device_hyper") or torch.cuda.is_current_stream_capturing():
        return False
    from .CLIP.clip.model import BatchNorm2d
    net = model.module if hasattr(model, "module") else model
    if not net.training:
        return False
    sync_bn = any(m.process_group is not None for m in net.modules() if isinstance(m, BatchNorm2d))
    if cfg.step_graph == "1":      # ONE graph: single-process steps only (collectives are never captured)
        return reducer is None and not sync_bn
    if sync_bn:                    # segmented replay: SyncBatchNorm through the mailbox transport only (device-side counter)
        from . import comm
        if any(m is None for m in comm.Mailbox._by_group.values()) or cfg.syncbn_comm != "mailbox":
            return False
    return True


def train_step(model, clip_model, optimizer, img, word_ids, neg_word_ids, args, lr_scheduler=None, reducer=None):
    """One optimisation step; returns the device tensor losses[4] (no host sync).

    TRIS_STEP_GRAPH=1: single-process steps are replayed from ONE hipGraph (tris_amd.graphs.GraphedTrainStep), captured at the
    first call for the batch shape seen there; other shapes (a ragged last batch), data-parallel runs and profiling passes run
    eagerly.  The returned tensor is then the graph's static output: read it (or clone it) before the next step.  Opt-in, because
    on ROCm 7.x it trades GPU time for host time (measured, B = 48, one box): a graph with parallel branches is launched node by
    node (host 37 ms, as eager) and its branches execute one after the other (50.0 ms/step against 45.0 eager on three streams);
    captured on ONE stream the runtime's packet path makes the launch 0.65 ms of host time per step, but the step is the
    single-stream step (50.1 ms) -- DESIGN.md section 5, "captured step"."""
    if _graphable(model, optimizer, img, reducer):
        net = model.module if hasattr(model, "module") else model
        key = (tuple(img.shape), tuple(word_ids.shape), None if neg_word_ids is None else tuple(neg_word_ids.shape),
               img.dtype, word_ids.dtype, id(clip_model), id(optimizer), id(lr_scheduler), id(reducer), ops.get_gemm_mode(),
               cfg.key())
        slot = net.__dict__.get("_tris_step_graph")
        if slot is not None and slot[0][5:] != key[5:]:
            # something the captured launches depend on changed for good (configuration, arithmetic, another optimiser / reducer /
            # aux model): what was recorded no longer describes the step -- record again.  (A different batch SHAPE alone, key[:5], is
            # the ragged last batch of an epoch: it runs eagerly below and the recording for the main shape is kept.)
            net.__dict__.pop("_tris_step_graph")
            slot = None
        if slot is not None and slot[1] is not None and slot[0] != key:
            # the same configuration, another batch shape.  Once in a while that is the ragged last batch of an epoch; several
            # steps in a row mean the recording was made on the odd one (a short first batch): record this shape in its place
            seen = net.__dict__.get("_tris_step_graph_other")
            n = seen[1] + 1 if seen is not None and seen[0] == key[:5] else 1
            net.__dict__["_tris_step_graph_other"] = (key[:5], n)
            if 0 < cfg.step_graph_rerecord <= n:
                net.__dict__.pop("_tris_step_graph")
                net.__dict__.pop("_tris_step_graph_other")
                net.__dict__.pop("_tris_step_graph_warned", None)
                slot = None
        elif slot is not None:
            net.__dict__.pop("_tris_step_graph_other", None)
        if slot is None:
            from .graphs import GraphedTrainStep, NotCapturable, SegmentedTrainStep
            cls = SegmentedTrainStep if cfg.step_graph == "seg" else GraphedTrainStep
            kw = {"reducer": reducer} if cls is SegmentedTrainStep else {}
            try:
                g = cls(model, clip_model, optimizer, args, (img, word_ids, neg_word_ids), lr_scheduler, **kw)
            except NotCapturable as e:
                import warnings
                warnings.warn(str(e))
                g = None
            slot = net.__dict__["_tris_step_graph"] = (key, g)
        if slot[0] == key and slot[1] is not None:
            return slot[1](img, word_ids, neg_word_ids)
        if slot[1] is not None and not net.__dict__.get("_tris_step_graph_warned"):
            import warnings
            net.__dict__["_tris_step_graph_warned"] = True
            warnings.warn(f"train_step: batch shape {tuple(img.shape)} differs from the captured one {slot[0][0]}: this step runs eagerly")
    losses = _step_body(model, clip_model, optimizer, img, word_ids, neg_word_ids, args, reducer)
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
        if idx % args.print_freq == 0 and args.distributed:
            # EVERY rank looks at its SyncBN mailbox time-out flag (and at the group's: one tiny all-reduce at the same point of
            # the loop on all ranks): a rank that gave up on an exchange has NaN statistics, and training must stop on all of them
            from . import comm
            comm.check_errors(collective=True)
        if idx % args.print_freq == 0 and ops.xattn_timed_out(collective=bool(args.distributed)):
            # a wait inside a persistent cross-attention launch gave up (a peer workgroup never became resident): that launch's
            # outputs are undefined -- stop instead of training on them (host sync: only where the loop prints)
            raise RuntimeError("fused cross attention: an in-kernel wait timed out; rerun with TRIS_XATTN_FUSED=0 (the two-launch pair)")
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


# ---- script level: python -m tris_amd.train_stage1 <reference flags>  (train_stage1.py:33-262, 413-437) --------------------
def setup_seed(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def _logger(args, rank):
    log = logging.getLogger("tris_amd")
    if not log.handlers:
        log.setLevel(logging.INFO if rank == 0 else logging.WARNING)
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("[%(asctime)s] %(message)s", "%H:%M:%S"))
        log.addHandler(h)
        if rank == 0 and getattr(args, "output", None):
            os.makedirs(args.output, exist_ok=True)
            log.addHandler(logging.FileHandler(os.path.join(args.output, f"log_rank{rank}.txt")))
    return log


def build_dataset(args, split, train, eval_mode, tokenizer=None):
    """ReferDataset with the reference's arguments (train_stage1.py:82-106, validate.py:50-62)"""
    from .dataset.ReferDataset import ReferDataset
    from .dataset.transform import get_transform
    return ReferDataset(refer_data_root=args.refer_data_root, dataset=args.dataset, splitBy=args.splitBy,
                        bert_tokenizer=args.bert_tokenizer, split=split, size=args.size, max_tokens=args.max_query_len,
                        image_transforms=get_transform(args.size, train=train), eval_mode=eval_mode,
                        negative_samples=args.negative_samples if not eval_mode else 0,
                        positive_samples=args.positive_samples, scales=getattr(args, "scales", False),
                        tokenizer=tokenizer)


def build_loader(args, dataset, batch_size, shuffle, distributed):
    """The HBM-resident loader (tris_amd.dataset.hbm) by default; TRIS_HBM_LOADER=0 selects the reference's
    DataLoader(num_workers=2, pin_memory=True) over the same dataset."""
    sampler = None
    if distributed:
        if getattr(dataset, "eval_mode", False):
            # no padding: DistributedSampler repeats refs to even out the shards, which would count them twice in the
            # all-reduced evaluation accumulators (tris_amd.validate)
            from .parallel import ShardSampler
            sampler = ShardSampler(dataset)
        else:
            # equal shard sizes on every rank (DistributedSampler pads), which SyncBatchNorm's count = M * world relies on
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(dataset, shuffle=shuffle)
    if cfg.hbm_loader:
        from .dataset.hbm import HbmLoader, HbmReferCache
        return HbmLoader(HbmReferCache(dataset, args.size), batch_size=batch_size, sampler=sampler,
                         shuffle=shuffle and sampler is None)
    from torch.utils.data import DataLoader
    return DataLoader(dataset, batch_size=batch_size, num_workers=2, pin_memory=True, sampler=sampler,
                      shuffle=shuffle and sampler is None)


def main(args, tokenizer=None):
    """Training driver with the reference's flow: build model / data / AdamW(2 groups) / poly LR, optional resume or
    --eval, then per epoch train_one_epoch -> validate on every test split -> keep the best-mIoU and best-hit checkpoints.
    The process-wide settings it chooses (step replay, the compute stream) are put back when it returns: a caller that goes on
    using the package in the same process -- the test suite -- finds them as it left them."""
    prev_graph = cfg.step_graph
    prev_stream = torch.cuda.current_stream() if torch.cuda.is_available() else None
    try:
        return _main(args, tokenizer)
    finally:
        cfg.step_graph = prev_graph
        if prev_stream is not None:
            torch.cuda.set_stream(prev_stream)


def _main(args, tokenizer=None):
    import torch.distributed as dist
    from .CLIP import clip
    from .model.model_stage1 import TRIS
    from .optim import FusedAdamW
    from .parallel import DataParallel, attach_reducer, convert_sync_batchnorm
    from .utils.util import load_checkpoint, load_pretrained_checkpoint, save_checkpoint
    from .validate import validate
    if args.distributed:
        if not dist.is_initialized():
            dist.init_process_group("nccl")
        local_rank = int(os.environ.get("LOCAL_RANK", dist.get_rank()))
        torch.cuda.set_device(local_rank)
        if dist.get_backend() == "nccl":
            cs = ops.place_streams()     # compute / text / weight-gradient streams on hardware queues the collective backend is not on
            if cs is not None:
                torch.cuda.set_stream(cs)
    else:
        local_rank = 0
        if cfg.own_stream and torch.cuda.is_available():
            torch.cuda.set_stream(ops.compute_stream())   # (never compute on the default stream: ops.compute_stream)
    if not cfg.step_graph_chosen:
        # (neither the environment nor the caller chose an issue form -- cfg.step_graph still holds its default)
        # the trainer replays its steps from the segmented hipGraphs (what bench.py reports as `value`): the host issues a step in
        # ~2 ms instead of ~30 ms of eager launches; batches of another shape (the last one of an epoch) run eagerly (train_step)
        cfg.step_graph = "seg"
    log = _logger(args, local_rank)
    net = TRIS(args).cuda(local_rank)
    param_groups = net.trainable_parameters()
    if args.distributed:
        convert_sync_batchnorm(net)
    model = DataParallel(net)
    log.info(f"number of params: {sum(p.numel() for p in net.parameters() if p.requires_grad) / 1e6: .2f}M")

    train_set = build_dataset(args, "train", train=True, eval_mode=args.eval, tokenizer=tokenizer)
    val_sets = [build_dataset(args, sp, train=False, eval_mode=True, tokenizer=tokenizer)
                for sp in args.test_split.split(",")]
    val_loaders = [build_loader(args, v, 1, False, args.distributed) for v in val_sets]

    optimizer = FusedAdamW([
        {"params": param_groups[0], "lr": args.lr * args.lr_multi, "weight_decay": args.weight_decay},
        {"params": param_groups[1], "lr": args.lr, "weight_decay": args.weight_decay},
    ], lr=args.lr, weight_decay=args.weight_decay)
    reducer = None
    if args.distributed:
        reducer = attach_reducer(net, optimizer)   # segmented all-reduce launched from inside backward

    def evaluate():
        res = [validate(args, vl, model, local_rank, logger=log) for vl in val_loaders]
        return res   # [(oIoU, mIoU, hit)] per split

    if args.resume and args.eval:
        if args.pretrain is not None:
            load_checkpoint(args, net, logger=log)
        t0 = time.time()
        res = evaluate()
        names = ("val", "testA", "testB")
        log.info(", ".join(f"{names[min(i, 2)]}: {float(r[1]):.4f}" for i, r in enumerate(res)))
        log.info(f"Testing time:  {datetime.timedelta(seconds=int(time.time() - t0))}")
        return res

    train_loader = build_loader(args, train_set, args.batch_size, True, args.distributed)
    steps_total = max(len(train_loader) * args.epoch, 1)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda x: (1 - x / steps_total) ** 0.9)
    if args.resume and args.pretrain is not None:
        load_checkpoint(args, net, optimizer, scheduler, log)

    log.info("Start training")
    clip_model, _ = clip.load("ViT-B-32", device=f"cuda:{local_rank}", jit=False, txt_length=args.max_query_len)
    freeze_aux(clip_model)
    best = {"val_acc": -1, "val_hit": -1, "epoch": -1, "path": "", "hit": -1, "hit_path": "", "testA": -1, "testB": -1}
    iteration, train_time, start = 0, 0.0, time.time()
    for epoch in range(args.start_epoch, args.epoch):
        t0 = time.time()
        if args.distributed and getattr(train_loader, "sampler", None) is not None:
            train_loader.sampler.set_epoch(epoch)
        iteration = train_one_epoch(train_loader, model, optimizer, epoch, local_rank, args, iteration, clip_model,
                                    lr_scheduler=scheduler, reducer=reducer, logger=log)
        torch.cuda.synchronize()
        train_time += time.time() - t0
        if args.distributed:
            # a SyncBatchNorm exchange that timed out poisoned its outputs: stop on every rank BEFORE validating / checkpointing
            from . import comm
            comm.check_errors(collective=True)
        if ops.xattn_timed_out(collective=bool(args.distributed)):    # (the sticky word: also covers the steps between two prints)
            raise RuntimeError("fused cross attention: an in-kernel wait timed out during the epoch; rerun with TRIS_XATTN_FUSED=0")
        res = evaluate()
        if ops.xattn_timed_out(collective=bool(args.distributed)):
            raise RuntimeError("fused cross attention: an in-kernel wait timed out during validation; rerun with TRIS_XATTN_FUSED=0")
        oIoU, val_acc, hit = res[0]
        if float(val_acc) > best["val_acc"] and local_rank == 0:
            if os.path.exists(best["path"]):
                os.remove(best["path"])
            best.update(path=save_checkpoint(epoch, net, optimizer, scheduler, log, args,
                                             f"ckpt_320_epoch_{epoch}_best.pth"),
                        val_acc=float(val_acc), val_hit=hit, epoch=epoch,
                        testA=float(res[1][1]) if len(res) > 1 else 0, testB=float(res[2][1]) if len(res) > 2 else 0)
        if hit > best["hit"] and local_rank == 0:
            if os.path.exists(best["hit_path"]):
                os.remove(best["hit_path"])
            best.update(hit_path=save_checkpoint(epoch, net, optimizer, scheduler, log, args,
                                                 f"ckpt_320_epoch_{epoch}_hit.pth"), hit=hit)
        if args.distributed:
            # rank 0 alone writes the checkpoints (up to two > 1 GB saves): its peers wait here instead of inside the next epoch's
            # first SyncBatchNorm exchange, whose spin is bounded (tris_amd.comm.Mailbox.SPIN_LIMIT)
            dist.barrier()
        log.info(str(best))
    if best["path"] and local_rank == 0:
        load_pretrained_checkpoint(best["path"], net)
    log.info(f"Training time {train_time:.1f}s; training + testing "
             f"{datetime.timedelta(seconds=int(time.time() - start))}")
    if args.distributed:
        from . import comm
        comm.check_errors(collective=True)
        comm.shutdown()   # unmap the peers' SyncBN mailboxes before the process group goes away
    return best


if __name__ == "__main__":
    from .args import get_parser
    _a = get_parser().parse_args()
    setup_seed(1234)
    main(_a)
