"""Data-parallel Stage-1 training over RCCL/xGMI: one process per GPU (reference: DistributedDataParallel +
SyncBatchNorm, train_stage1.py:69-70, 435-437).

The only exchange steps of the path are (1) the gradient mean over ranks and (2) SyncBatchNorm statistics.
(1) runs on the flat gradient arenas of tris_amd.optim in a few large chunks (xGMI rings are per-link bound, so
big messages; ~400 MB total) launched from inside backward; (2) lives in ops.BatchNormFn (all_gather of
[mean|var|count], all_reduce of the two backward sums).  Nothing else crosses ranks: the in-batch contrastive heads
are rank-local (model_stage1.py:66,107).  Works with any torch.distributed backend through tris_amd.comm ("nccl" =
RCCL on ROCm; "gloo" for the CPU tests of the host logic and the two-ranks-on-one-GPU parity test).

Which segment may be released WHERE (the invariant the reducer rests on): autograd's engine runs, among the nodes
that are ready, the one created LAST in forward first.  A `boundary()` placed in the forward graph therefore runs its
backward only after every node created after it has run -- i.e. after every gradient produced "downstream" of that
point has been written.  WHERE TRIS.forward issues the text encoder (before the trunk, in the middle of it, behind it:
TRIS_TEXT_AT, default behind layer4) decides where its backward runs relative to the trunk's, so its parameters must not
ride on a trunk boundary.  They have their own boundaries on the text path (behind blocks 7 and 3 and behind the
embedding), and the embedding tables, whose gradient is written after the last of them, are released by `finish()`.
"""


import torch
import torch.distributed as dist

from . import comm
from .config import cfg
from .CLIP.clip.model import BatchNorm2d


def convert_sync_batchnorm(module, process_group=None):
    """Equivalent of nn.SyncBatchNorm.convert_sync_batchnorm for tris_amd BatchNorm2d layers."""
    group = process_group if process_group is not None else dist.group.WORLD
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            m.process_group = group
    return module


class ReducerOrderError(RuntimeError):
    pass


class GradReducer:
    """Mean-all-reduce of the flat gradient arenas, overlapped with backward.

    The arenas are cut into SEGMENTS in the order backward completes them.  `boundary(x, k)` is an identity placed in
    the forward graph at a point where segment k's last gradient has been written once backward passes it; its backward
    launches the (asynchronous, chunked) all-reduce of that segment while the rest of backward keeps computing.
    `finish()` before the optimiser step launches whatever is left and waits.  With one rank everything is a no-op
    (unless force=True, which runs the collectives for code-path testing).

    check=True (env TRIS_DDP_CHECK=1; tests): `poison()` fills every parameter's gradient with NaN before backward and
    every launch first verifies -- with a device sync -- that its segment holds no NaN any more, i.e. that no segment is
    released before its last producer ran.  Raises ReducerOrderError otherwise.
    """

    def __init__(self, flats, group=None, chunk_mb=64, force=False, check=None):
        self.flats = list(flats)
        self.group = group
        self.chunk = chunk_mb * (1 << 20) // 4
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force
        self.segments = {}      # key -> list of (flat index, start, end)           : what is all-reduced
        self.param_ranges = {}  # key -> list of (flat index, start, end, name)     : the parameters inside (no padding)
        self.pending = []
        self.done = set()
        self.launch_log = []    # keys in launch order of the current / last step (tests)
        self.active = self.world > 1 or force
        self.check = cfg.ddp_check if check is None else check
        # NCCL/RCCL averages in the collective; gloo has no AVG: sum, then scale in finish()
        self.prescaled = []
        self.avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        # Sparse exchange of the token-embedding gradient (TRIS_DDP_SPARSE_EMBED=0 switches back to the dense all-reduce): the
        # table is [49408, 512] fp32 = 101 MB, of which a rank touches <= B*L rows per step.  `sparse_exclude` lists the arena
        # ranges left out of the dense segments; take_embedding_rows() does the exchange.
        self.sparse_embed = cfg.ddp_sparse_embed
        self.sparse_exclude = {}   # id(param) -> (flat index, start, end)
        self.sparse_log = []       # (rows gathered, bytes on the wire per rank) of the current / last step (tests, dist_check)
        self.exposed = None        # (event before, event after) the compute stream's wait in finish(): exposed communication

    # ---- segment planning ------------------------------------------------------------------------------------------
    @staticmethod
    def plan(arenas, named_params, rules, with_params=False):
        """rules: ordered {segment key: predicate(param name)}.  Returns {key: [(arena idx, start, end), ...]} covering
        every arena slot exactly once (parameters matching no rule fall into the LAST key).  with_params=True also
        returns {key: [(arena idx, start, end, name)]} with the exact extent of each parameter."""
        name_of = {id(p): n for n, p in named_params}
        keys = list(rules)
        seg = {k: [] for k in keys}
        par = {k: [] for k in keys}
        for ai, ar in enumerate(arenas):
            cur_key, cur_start = None, 0
            for p, o in zip(ar.params, ar.offsets):
                n = name_of.get(id(p), "")
                k = next((kk for kk in keys if rules[kk](n)), keys[-1])
                # (exact extents feed the NaN-poison order check.  A parameter whose gradient arrives through autograd's
                # AccumulateGrad -- `_tris_accumulates`: the ViT trunk's class / positional embedding -- ADDS into its arena
                # slot, which zero_grad() cleared: poisoning it would undo the zeroing and always trip the check, so it is
                # left out of the check, not of the segment)
                if not getattr(p, "_tris_accumulates", False):
                    par[k].append((ai, o, o + p.numel(), n))
                if k != cur_key:
                    if cur_key is not None:
                        seg[cur_key].append((ai, cur_start, o))
                    cur_key, cur_start = k, o
            if cur_key is not None:
                seg[cur_key].append((ai, cur_start, ar.numel))
        return (seg, par) if with_params else seg

    def set_segments(self, segments, param_ranges=None):
        self.segments = segments
        self.param_ranges = param_ranges or {}

    # ---- order check (tests / TRIS_DDP_CHECK=1) -----------------------------------------------------------------------
    def poison(self):
        """NaN into every parameter's gradient (not the padding): call before backward when check is on"""
        for ranges in self.param_ranges.values():
            for ai, s, e, _ in ranges:
                self.flats[ai][s:e].fill_(float("nan"))

    def _verify(self, key):
        from . import ops
        ops.wgrad_join()
        if self.flats and self.flats[0].is_cuda:
            torch.cuda.synchronize()
        bad = [n for ai, s, e, n in self.param_ranges.get(key, []) if bool(torch.isnan(self.flats[ai][s:e]).any())]
        if bad:
            raise ReducerOrderError(f"segment {key!r} released before the gradient of {bad[:4]} "
                                    f"(+{max(len(bad) - 4, 0)} more) was written")

    # ---- runtime -----------------------------------------------------------------------------------------------------
    def _launch(self, key):
        """called by the boundary nodes from inside backward.  While a segmented capture of the step is being recorded
        (tris_amd.graphs.SegmentedTrainStep) nothing is launched: the capture notes WHERE the segment was released and the
        replay issues the collectives there, between two graph replays -- collectives are never part of a graph."""
        from . import ops
        if ops._SEG is not None:
            if self.active:
                ops._SEG.release(key)
            return
        self._launch_now(key)

    def _launch_now(self, key):
        if key in self.done or not self.active:
            return
        self.done.add(key)
        self.launch_log.append(key)
        from . import ops
        if self.check and self.param_ranges:
            self._verify(key)
        # The collectives are issued from the reducer's OWN stream: it (not the compute stream) waits for the weight-gradient
        # and text streams that may still be writing this segment, so backward keeps running while the segment drains; the
        # compute stream meets the collectives again only in finish().
        on_gpu = bool(self.flats) and self.flats[0].is_cuda
        if on_gpu:
            rs = ops.side_stream("reduce")
            rs.wait_stream(torch.cuda.current_stream())
            ops.wgrad_join(into=rs)
            ctx = torch.cuda.stream(rs)
        else:
            import contextlib
            ops.wgrad_join()
            ctx = contextlib.nullcontext()
        with ctx:
            for ai, s, e in self._dense_ranges(key):
                f = self.flats[ai]
                for c in range(s, e, self.chunk):
                    self.pending.append(comm.all_reduce(f[c:min(e, c + self.chunk)], op=self.op, group=self.group,
                                                        async_op=True))

    def _dense_ranges(self, key):
        """the ranges of segment `key` that are all-reduced densely: everything except the sparsely exchanged tables"""
        out = []
        for ai, s, e in self.segments.get(key, []):
            cuts = sorted((xs, xe) for xa, xs, xe in self.sparse_exclude.values() if xa == ai and xs >= s and xe <= e)
            cur = s
            for xs, xe in cuts:
                if xs > cur:
                    out.append((ai, cur, xs))
                cur = max(cur, xe)
            if cur < e:
                out.append((ai, cur, e))
        return out

    def exclude_sparse(self, param, ai, start, end):
        self.sparse_exclude[id(param)] = (ai, start, end)

    def take_embedding_rows(self, param, ids, rows, dtok):
        """Called from the embedding's own backward (ops.EmbedFn) with this rank's row list: all-gather (ids, rows) of every
        rank and build dtok = (1 / world) * scatter-sum of ALL rows with the deterministic kernel -- the same input in the same
        order on every rank, hence bit-identical gradients.  Returns False when the table is not exchanged sparsely (the
        caller then builds its local gradient and the dense all-reduce of the segment carries it)."""
        if not (self.active and self.sparse_embed and id(param) in self.sparse_exclude):
            return False
        from . import ops
        if ops._SEG is not None:      # segmented capture: the exchange is issued by the replay, behind the text backward graph
            ops._SEG.embed_rows = (ids, rows, dtok)
            return True
        self.exchange_rows(ids, rows, dtok)
        return True

    def exchange_rows(self, ids, rows, dtok):
        from . import ops
        R, W = rows.shape
        world = max(self.world, 1)
        # (backends without AVG -- gloo -- scale the whole arenas by 1 / world in finish(): the table must not be scaled twice)
        scale = 1.0 / world if self.avg else 1.0
        if self.world > 1:
            all_ids = torch.empty(world * R, device=ids.device, dtype=ids.dtype)
            all_rows = torch.empty(world * R, W, device=rows.device, dtype=rows.dtype)
            on_gpu = rows.is_cuda
            if on_gpu:   # issued from the reducer's stream like every other collective of the step
                rs = ops.side_stream("reduce")
                rs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(rs):
                    comm.all_gather_into(all_ids, ids.contiguous(), group=self.group)
                    comm.all_gather_into(all_rows.view(-1), rows.contiguous().view(-1), group=self.group)
                    ops.call("tris_embed_rows_bwd_f32", ops.P(all_ids), ops.P(all_rows), ops.P(dtok), world * R, W, scale,
                             rs.cuda_stream)
                for t in (ids, rows, all_ids, all_rows, dtok):
                    t.record_stream(rs)
            else:
                comm.all_gather_into(all_ids, ids.contiguous(), group=self.group)
                comm.all_gather_into(all_rows.view(-1), rows.contiguous().view(-1), group=self.group)
                dtok.index_add_(0, all_ids, all_rows * scale)
        else:   # one rank (force=True): the same kernel on the own list
            if rows.is_cuda:
                ops.call("tris_embed_rows_bwd_f32", ops.P(ids), ops.P(rows), ops.P(dtok), R, W, 1.0, ops._stream())
            else:
                dtok.index_add_(0, ids, rows)
        self.sparse_log.append((world * R, R * (W * 4 + 8)))
        self.last_rows = (ids, rows)

    def boundary(self, x, key):
        if not self.active or not torch.is_grad_enabled() or not x.requires_grad:
            return x
        y = _Boundary.apply(x, self, key)
        for attr in ("_bn_link", "_pl", "_h2"):   # (ops._BnBwdLink / the plane tag ride on the tensor: the boundary is an identity)
            v = getattr(x, attr, None)
            if v is not None:
                setattr(y, attr, v)
        return y

    def finish(self):
        """Launch every segment not yet launched, wait for all, reset for the next step."""
        if self.active:
            if self.segments:
                for k in self.segments:
                    self._launch_now(k)
            else:  # no plan: whole arenas
                from . import ops
                ops.wgrad_join()
                for f in self.flats:
                    for c in range(0, f.numel(), self.chunk):
                        self.pending.append(comm.all_reduce(f[c:c + self.chunk], op=self.op, group=self.group,
                                                            async_op=True))
            on_gpu = bool(self.flats) and self.flats[0].is_cuda
            if on_gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()    # from here the compute stream only waits: what it waits for is exposed communication
            for h in self.pending:
                h.wait()
            if on_gpu:
                from . import ops
                torch.cuda.current_stream().wait_stream(ops.side_stream("reduce"))
                e1.record()
                self.exposed = (e0, e1)
            if not self.avg and self.world > 1:
                for ai, f in enumerate(self.flats):    # (what scale_now() already divided is left alone)
                    cur = 0
                    for s, e in sorted((s, e) for a_, s, e in self.prescaled if a_ == ai):
                        if s > cur:
                            f[cur:s].mul_(1.0 / self.world)
                        cur = max(cur, e)
                    if cur < f.numel():
                        f[cur:].mul_(1.0 / self.world)
        self.pending = []
        self.prescaled = []
        self.done = set()

    def scale_now(self, key):
        """SUM backends: divide segment `key` by the world size on the current stream (its all-reduce has completed there) instead of
        in finish() -- the replayed step updates the segment's parameters right behind this (graphs.SegmentedTrainStep)"""
        if self.avg or self.world <= 1:
            return
        for ai, s, e in self._dense_ranges(key):
            self.flats[ai][s:e].mul_(1.0 / self.world)
            self.prescaled.append((ai, s, e))

    def exposed_ms(self):
        """milliseconds the compute stream waited for collectives in the last finish() (host sync; None before the first step)"""
        if self.exposed is None:
            return None
        self.exposed[1].synchronize()
        return self.exposed[0].elapsed_time(self.exposed[1])

    def begin_step(self):
        self.launch_log = []
        self.sparse_log = []
        if self.check and self.active:
            self.poison()

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


# where each segment is released (tris_amd.CLIP.clip.model places the boundaries):
#   "heads"  : vis_project / lan_project / attn_fusion           -- boundary behind layer4 (created before every head node)
#   "layer4" ... "layer1" : boundary behind the previous stage
#   "text_hi" / "text_mid" / "text" : text transformer blocks 8-11 (+ ln_final, text_projection) / 4-7 / 0-3 -- boundaries
#              on the text path behind blocks 7 and 3 and behind the embedding.  The text encoder's backward is the LAST part
#              of backward (it is issued first in forward), so whatever of it is still un-reduced when backward ends is
#              exposed communication: three sub-segments leave only the last third (+ the embedding) to finish()
#   "embed"  : token / positional embedding (written after the "text" boundary) and
#   "stem"   : the three stem convolutions (last to finish in the trunk) -- finish()
def _text_block(name):
    """index of the text-transformer block a parameter belongs to; ln_final / text_projection count as behind the last
    block (99); anything else -1"""
    pre = "backbone.transformer.resblocks."
    if name.startswith(pre):
        return int(name[len(pre):].split(".")[0])
    if name.startswith("backbone.ln_final.") or name == "backbone.text_projection":
        return 99
    return -1


TEXT_BOUNDARIES = {7: "text_hi", 3: "text_mid"}   # behind block i: every block > i (created later) has finished its backward


STAGE1_RULES = {
    "heads": lambda n: not n.startswith("backbone."),
    "embed": lambda n: n in ("backbone.token_embedding.weight", "backbone.positional_embedding"),
    "text_hi": lambda n: (not n.startswith("backbone.visual.")) and _text_block(n) >= 8,
    "text_mid": lambda n: (not n.startswith("backbone.visual.")) and 4 <= _text_block(n) < 8,
    "text": lambda n: not n.startswith("backbone.visual."),
    "layer4": lambda n: n.startswith("backbone.visual.layer4."),
    "layer3": lambda n: n.startswith("backbone.visual.layer3."),
    "layer2": lambda n: n.startswith("backbone.visual.layer2."),
    "layer1": lambda n: n.startswith("backbone.visual.layer1."),
    "stem": lambda n: True,
}


def stage1_segments(model, optimizer, with_params=False):
    """Segments of the Stage-1 graph (tris_amd.model.model_stage1.TRIS.forward) in the order backward may release them."""
    return GradReducer.plan(optimizer.arenas, list(model.named_parameters()), STAGE1_RULES, with_params)


def attach_reducer(model, optimizer, group=None, force=False, check=None, chunk_mb=64):
    """GradReducer over the optimiser's gradient arenas, planned for `model` (a TRIS) and hooked into its forward."""
    net = model.module if hasattr(model, "module") else model
    red = GradReducer([a.g for a in optimizer.arenas], group=group, chunk_mb=chunk_mb, force=force, check=check)
    seg, par = stage1_segments(net, optimizer, with_params=True)
    red.set_segments(seg, par)
    tok = getattr(getattr(net, "backbone", None), "token_embedding", None)
    if tok is not None and red.sparse_embed:   # the token table travels as rows (take_embedding_rows), not densely
        for ai, ar in enumerate(optimizer.arenas):
            for p, o in zip(ar.params, ar.offsets):
                if p is tok.weight:
                    red.exclude_sparse(p, ai, o, o + (p.numel() + 63) // 64 * 64)
    net.backbone.visual.grad_reducer = red     # trunk boundaries (ModifiedResNet.forward_cl)
    net.backbone.grad_reducer = red            # text boundary (CLIP.encode_text)
    return red


class DataParallel(torch.nn.Module):
    """`model.module` wrapper expected by the reference's callers (train_stage1.py:74); broadcasts rank-0 weights
    and buffers at construction like DDP does."""

    def __init__(self, module, group=None):
        super().__init__()
        self.module = module
        self.group = group
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                comm.broadcast(t.data, src=0, group=group)

    def forward(self, *a, **k):
        return self.module(*a, **k)


class ShardSampler:
    """Evaluation sampler: rank r takes dataset indices r, r + world, ...  Unlike DistributedSampler it does NOT pad to
    a multiple of the world size, so no ref is counted twice when the five evaluation accumulators are all-reduced
    (validate.py:237-249 reports rank-local meters; tris_amd.validate returns the global numbers)."""

    def __init__(self, dataset, rank=None, world=None):
        self.n = len(dataset)
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world

    def __iter__(self):
        return iter(range(self.rank, self.n, self.world))

    def __len__(self):
        return len(range(self.rank, self.n, self.world))

    def set_epoch(self, epoch):
        pass
