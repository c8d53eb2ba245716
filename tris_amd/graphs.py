"""hipGraph capture of the Stage-1 inference path (validate.py hot loop at batch 1 is launch-latency bound:
~450 short kernels per forward).  Two graphs, matching the loop structure of `validate`:

    visual(img)      RN50 trunk -> vis_project -> L2 norm        replayed once per image
    sentence(ids)    text encoder -> cross attention -> maps     replayed once per sentence of that image

Inputs are copied into static buffers; outputs are static buffers owned by the graphs (consume or clone them before
the next replay).  Capture uses torch's stream-capture plumbing (torch.cuda.graph); every captured node is one of
this repo's HIP kernels launched through the C ABI on the capturing stream."""
import torch


class GraphedStage1Eval:
    def __init__(self, net, img_shape, query_len, warmup=2):
        assert not net.training, "capture the eval path (model.eval())"
        self.net = net
        dev = next(net.parameters()).device
        self.s_img = torch.zeros(img_shape, device=dev, dtype=torch.float32)
        self.s_ids = torch.zeros(img_shape[0], query_len, device=dev, dtype=torch.int64)
        self.s_ids[:, 0] = 49406
        self.s_ids[:, 1] = 49407
        H = img_shape[2]
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):  # first-call setup (hipFuncSetAttribute, workspace growth) outside capture
                    v = net.encode_visual(self.s_img)
                    net.forward_cached(v, self.s_ids, H)
            torch.cuda.current_stream().wait_stream(side)
            self.g_vis = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_vis):
                self.vis = net.encode_visual(self.s_img)
            self.g_txt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_txt):
                self.out = net.forward_cached(self.vis, self.s_ids, H)

    def visual(self, img):
        self.s_img.copy_(img, non_blocking=True)
        self.g_vis.replay()

    def sentence(self, ids):
        self.s_ids.copy_(ids, non_blocking=True)
        self.g_txt.replay()
        return self.out


class GraphedFrozenText:
    """hipGraph replay of a FROZEN text tower (the aux CLIP of the Stage-1 loss, train_stage1.py:167, 346-347): ~130 short
    kernels per step whose only cost is the host time to issue them -- time during which the compute stream has nothing
    queued.  Static shapes ([n sentences, L tokens]), no gradient: captured once (after two eager warm-up calls, so that the
    GEMM autotuner has seen every shape), replayed as ONE launch per step on the caller's side stream.  The output is a static
    buffer: consume it before the next replay (the step's own stream order guarantees that)."""

    def __init__(self, clip_model, n, L, warmup=2):
        dev = next(clip_model.parameters()).device
        self.key = (n, L, clip_model.token_embedding.weight.data_ptr())
        self.s_ids = torch.zeros(n, L, device=dev, dtype=torch.int64)
        self.s_ids[:, 0] = 49406
        self.s_ids[:, 1] = 49407
        with torch.no_grad():
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                for _ in range(warmup):
                    clip_model.encode_text(self.s_ids)
            torch.cuda.current_stream().wait_stream(cap)
            self.g = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (the collective backend's watchdog, the autograd engine of a previous
            # step) may touch the device while this stream is capturing
            with torch.cuda.graph(self.g, capture_error_mode="thread_local"):
                self.out = clip_model.encode_text(self.s_ids)[1]

    def __call__(self, ids):
        """replay on the CURRENT stream; returns a COPY of the static [n, E] output (a few hundred KB): the loss keeps it for
        its backward, and a second forward before that backward (gradient accumulation, an evaluation helper reusing the step's
        forward) would otherwise overwrite a tensor autograd still holds"""
        self.s_ids.copy_(ids, non_blocking=True)
        self.g.replay()
        return self.out.clone()


class GraphedTrainStep:
    """The WHOLE Stage-1 training step (TRIS forward, loss block with the frozen aux CLIP, backward, AdamW) as ONE hipGraph.

    Why: the eager step is ~1 500 ctypes launches issued from Python / the autograd engine -- 38 ms of host work against a
    43 ms GPU step (BENCH_r02: host_issue_ms_per_step) -- so the next kernel gains would run into the launch path.  Replayed,
    the host's share of a step is one input copy, twelve bytes of optimiser scalars and one graph launch.

    How: torch's stream capture around the very code the eager step runs (train_stage1._step_body): every node is one of this
    repo's kernels launched through the C ABI.  The text and weight-gradient side streams fork from / join the capturing stream
    through events, so the graph keeps the three-branch overlap of the eager step.  What a graph freezes and the step changes:
      * inputs          -> static buffers, copied into before each replay;
      * lr, Adam bias corrections -> a device tensor refreshed by the host per replay (FusedAdamW.push_hyper);
      * BatchNorm step counters   -> host-side, bumped per replay.
    Priming (first-encounter GEMM autotuning times kernels with HIP events, workspaces and allocator pools grow: none of that
    may happen under capture) is two eager forward + backward passes WITHOUT an optimiser step, with the BatchNorm running
    statistics saved and restored around them -- so the captured step starts from exactly the state the caller handed over and
    step k of a replayed run equals step k of an eager run.

    Not captured (the eager step runs instead): data-parallel runs (the SyncBatchNorm mailboxes carry a host-side parity
    and the reducer issues collectives from autograd hooks), profiling passes, batches of another shape."""

    def __init__(self, model, clip_model, optimizer, args, example, lr_scheduler=None, priming=2):
        from . import ops
        from .CLIP.clip.model import BatchNorm2d
        from .train_stage1 import _step_body
        img, ids, neg = example
        self.model, self.clip_model, self.optimizer, self.args, self.sched = model, clip_model, optimizer, args, lr_scheduler
        self.s_img = img.detach().clone()
        self.s_ids = ids.detach().clone()
        self.s_neg = None if neg is None else neg.detach().clone()
        net = model.module if hasattr(model, "module") else model
        self.bns = [m for m in net.modules() if isinstance(m, BatchNorm2d)]
        optimizer.enable_device_hyper()
        # ---- priming: eager forward + backward, no optimiser step, BatchNorm running statistics put back afterwards
        keep = [(m.running_mean.clone(), m.running_var.clone(), m._nbt_pending) for m in self.bns]
        for _ in range(priming):
            _step_body(model, clip_model, optimizer, self.s_img, self.s_ids, self.s_neg, args, None, optimizer_step=False)
        torch.cuda.synchronize()
        with torch.no_grad():
            for m, (rm, rv, nbt) in zip(self.bns, keep):
                m.running_mean.copy_(rm)
                m.running_var.copy_(rv)
                m._nbt_pending = nbt
        del keep
        torch.cuda.synchronize()
        torch.cuda.empty_cache()   # the priming passes' activation pool is not needed again (the graph has its own)
        nbt = [m._nbt_pending for m in self.bns]
        self.g = torch.cuda.CUDAGraph()
        ops._CAPTURE_STREAMS = True
        try:
            with torch.cuda.graph(self.g, capture_error_mode="thread_local"):
                self.losses = _step_body(model, clip_model, optimizer, self.s_img, self.s_ids, self.s_neg, args, None,
                                         device_hyper=True)
        finally:
            ops._CAPTURE_STREAMS = False
        for m, n in zip(self.bns, nbt):   # (capture launched nothing: the forward's host-side count is taken back)
            m._nbt_pending = n

    def __call__(self, img, ids, neg):
        self.s_img.copy_(img, non_blocking=True)
        self.s_ids.copy_(ids, non_blocking=True)
        if self.s_neg is not None:
            self.s_neg.copy_(neg, non_blocking=True)
        opt = self.optimizer
        opt._steps += 1
        opt.push_hyper()
        self.g.replay()
        for m in self.bns:
            m._nbt_pending += 1
        if self.sched is not None:
            self.sched.step()
        return self.losses


def frozen_text(clip_model, ids):
    """encode_text(ids)[1] of a frozen tower through a cached hipGraph (TRIS_HIPGRAPH=0: eager)"""
    import os
    if os.environ.get("TRIS_HIPGRAPH", "1") == "0" or torch.cuda.is_current_stream_capturing() or torch.is_grad_enabled():
        return clip_model.encode_text(ids)[1]
    n, L = ids.shape
    cache = clip_model.__dict__.setdefault("_tris_text_graphs", {})
    key = (n, L, clip_model.token_embedding.weight.data_ptr())
    g = cache.get(key)
    if g is None:
        if len(cache) >= 4:
            cache.clear()
        g = cache[key] = GraphedFrozenText(clip_model, n, L)
    return g(ids)
