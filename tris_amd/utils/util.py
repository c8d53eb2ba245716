"""Host utilities with the reference's names and contracts (utils/util.py): metric helper, meters and the
checkpoint format {'model','optimizer','lr_scheduler','epoch'} (:50-107)."""
import os

import torch


def compute_mask_IU(masks, target):
    """(I, U) pixel counts of two boolean masks of equal spatial size (utils/util.py:9-15)."""
    if target.shape[-2:] != masks.shape[-2:]:
        raise ValueError(f"mask sizes differ: {tuple(masks.shape)} vs {tuple(target.shape)}")
    return torch.sum(torch.logical_and(masks, target)), torch.sum(torch.logical_or(masks, target))


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _state(x):
    return x.state_dict() if hasattr(x, "state_dict") else x


def save_checkpoint(epoch, model, optimizer, lr_schdeduler, logger=None, args=None, checkpoint_name=None):
    state = {"model": model.state_dict(), "optimizer": _state(optimizer), "lr_scheduler": _state(lr_schdeduler),
             "epoch": epoch}
    os.makedirs(args.output, exist_ok=True)
    path = os.path.join(args.output, checkpoint_name or f"ckpt_448_epoch_{epoch}.pth")
    torch.save(state, path)
    (logger.info if logger is not None else print)(f"{path} saved !!!")
    return path


def load_checkpoint(args, model_without_ddp, optimizer=None, lr_scheduler=None, logger=None):
    path = os.path.join(args.output, args.pretrain)
    ckpt = torch.load(path, map_location="cpu")
    model_without_ddp.load_state_dict(ckpt["model"], strict=False)
    if not args.eval and all(k in ckpt for k in ("optimizer", "lr_scheduler", "epoch")) and optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
        if lr_scheduler is not None:
            lr_scheduler.load_state_dict(ckpt["lr_scheduler"])
        args.start_epoch = ckpt["epoch"] + 1
    (logger.info if logger is not None else print)(f"=> loaded successfully '{args.pretrain}'")


def load_pretrained_checkpoint(name, model_without_ddp):
    ckpt = torch.load(name, map_location="cpu")
    print(model_without_ddp.load_state_dict(ckpt["model"], strict=False))
