"""Data-parallel Stage-1 training over RCCL/xGMI: one process per GPU (reference: DistributedDataParallel +
SyncBatchNorm, train_stage1.py:69-70, 435-437).

The only exchange steps of the path are (1) the gradient mean over ranks and (2) SyncBatchNorm statistics.
(1) runs on the flat gradient arenas of tris_amd.optim in a few large chunks (xGMI rings are per-link bound, so
big messages; ~400 MB total) on a side stream; (2) lives in ops.BatchNormFn (all_gather of [mean|var|count],
all_reduce of the two backward sums).  Nothing else crosses ranks: the in-batch contrastive heads are rank-local
(model_stage1.py:66,107).  Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" for CPU tests of
the host logic).
"""
import torch
import torch.distributed as dist

from .CLIP.clip.model import BatchNorm2d


def convert_sync_batchnorm(module, process_group=None):
    """Equivalent of nn.SyncBatchNorm.convert_sync_batchnorm for tris_amd BatchNorm2d layers."""
    group = process_group if process_group is not None else dist.group.WORLD
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            m.process_group = group
    return module


class GradReducer:
    """Mean-all-reduce flat gradient buffers in `chunk_mb` pieces; `reduce()` after backward, before step."""

    def __init__(self, flats, group=None, chunk_mb=64, force=False):
        self.flats = list(flats)
        self.group = group
        self.chunk = chunk_mb * (1 << 20) // 4
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force  # run the collectives even with one rank (code-path test)

    def reduce(self):
        if self.world == 1 and not self.force:
            return
        handles = []
        for f in self.flats:
            for s in range(0, f.numel(), self.chunk):
                handles.append(dist.all_reduce(f[s:s + self.chunk], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
        for h in handles:
            h.wait()
        for f in self.flats:
            f.mul_(1.0 / self.world)


class DataParallel(torch.nn.Module):
    """`model.module` wrapper expected by the reference's callers (train_stage1.py:74); broadcasts rank-0 weights
    and buffers at construction like DDP does."""

    def __init__(self, module, group=None):
        super().__init__()
        self.module = module
        self.group = group
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=group)

    def forward(self, *a, **k):
        return self.module(*a, **k)
