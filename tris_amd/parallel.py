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
    """Mean-all-reduce of the flat gradient arenas, overlapped with backward.

    The arenas are cut into SEGMENTS in the order backward completes them (heads + text encoder first, then the RN50
    trunk from layer4 down to the stem).  `boundary(x, k)` is an identity placed in the forward graph where segment k's
    last gradient has been written once backward passes it; its backward launches the (asynchronous, chunked)
    all-reduce of that segment on RCCL's stream while the rest of backward keeps computing.  `finish()` before the
    optimiser step launches whatever is left and waits.  With one rank everything is a no-op (unless force=True,
    which runs the collectives for code-path testing).
    """

    def __init__(self, flats, group=None, chunk_mb=64, force=False):
        self.flats = list(flats)
        self.group = group
        self.chunk = chunk_mb * (1 << 20) // 4
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force
        self.segments = {}      # key -> list of (flat index, start, end)
        self.pending = []
        self.done = set()
        self.active = self.world > 1 or force
        # NCCL/RCCL averages in the collective; gloo (CPU tests) has no AVG: sum, then scale in finish()
        self.avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM

    # ---- segment planning ------------------------------------------------------------------------------------------
    @staticmethod
    def plan(arenas, named_params, rules):
        """rules: ordered {segment key: predicate(param name)}.  Returns {key: [(arena idx, start, end), ...]} covering
        every arena slot exactly once (parameters matching no rule fall into the LAST key)."""
        name_of = {id(p): n for n, p in named_params}
        keys = list(rules)
        seg = {k: [] for k in keys}
        for ai, ar in enumerate(arenas):
            ends = ar.offsets[1:] + [ar.numel]
            cur_key, cur_start = None, 0
            for p, o, e in zip(ar.params, ar.offsets, ends):
                n = name_of.get(id(p), "")
                k = next((kk for kk in keys if rules[kk](n)), keys[-1])
                if k != cur_key:
                    if cur_key is not None:
                        seg[cur_key].append((ai, cur_start, o))
                    cur_key, cur_start = k, o
            if cur_key is not None:
                seg[cur_key].append((ai, cur_start, ar.numel))
        return seg

    def set_segments(self, segments):
        self.segments = segments

    # ---- runtime -----------------------------------------------------------------------------------------------------
    def _launch(self, key):
        if key in self.done or not self.active:
            return
        self.done.add(key)
        from . import ops
        ops.wgrad_join()  # the segment's weight gradients may still be in flight on the weight-gradient stream
        for ai, s, e in self.segments.get(key, []):
            f = self.flats[ai]
            for c in range(s, e, self.chunk):
                self.pending.append(dist.all_reduce(f[c:min(e, c + self.chunk)], op=self.op, group=self.group,
                                                    async_op=True))

    def boundary(self, x, key):
        if not self.active or not torch.is_grad_enabled():
            return x
        return _Boundary.apply(x, self, key)

    def finish(self):
        """Launch every segment not yet launched, wait for all, reset for the next step."""
        if self.active:
            if self.segments:
                for k in self.segments:
                    self._launch(k)
            else:  # no plan: whole arenas
                for f in self.flats:
                    for c in range(0, f.numel(), self.chunk):
                        self.pending.append(dist.all_reduce(f[c:c + self.chunk], op=self.op, group=self.group,
                                                            async_op=True))
            for h in self.pending:
                h.wait()
            if not self.avg and self.world > 1:
                for f in self.flats:
                    f.mul_(1.0 / self.world)
        self.pending = []
        self.done = set()

    reduce = finish  # the non-overlapped entry point keeps working


class _Boundary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, reducer, key):
        ctx.reducer, ctx.key = reducer, key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.reducer._launch(ctx.key)
        return g, None, None


def stage1_segments(model, optimizer):
    """Backward completion order of the Stage-1 graph (tris_amd.model.model_stage1.TRIS.forward): heads and the text
    encoder finish first (they are created last in forward), then the trunk from layer4 down to the stem."""
    rules = {
        "heads_text": lambda n: not n.startswith("backbone.visual."),
        "layer4": lambda n: n.startswith("backbone.visual.layer4."),
        "layer3": lambda n: n.startswith("backbone.visual.layer3."),
        "layer2": lambda n: n.startswith("backbone.visual.layer2."),
        "layer1": lambda n: n.startswith("backbone.visual.layer1."),
        "stem": lambda n: True,
    }
    return GradReducer.plan(optimizer.arenas, list(model.named_parameters()), rules)


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
