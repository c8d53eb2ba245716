"""hipGraph capture of the Stage-1 inference path (validate.py hot loop at batch 1 is launch-latency bound:
~450 short kernels per forward).  Two graphs, matching the loop structure of `validate`:

    visual(img)      RN50 trunk -> vis_project -> L2 norm        replayed once per image
    sentence(ids)    text encoder -> cross attention -> maps     replayed once per sentence of that image

Inputs are copied into static buffers; outputs are static buffers owned by the graphs (consume or clone them before
the next replay).  Capture uses torch's stream-capture plumbing (torch.cuda.graph); every captured node is one of
this repo's HIP kernels launched through the C ABI on the capturing stream."""


import torch


class NotCapturable(RuntimeError):
    """the step cannot be replayed from hipGraphs in this configuration (train_step then runs it eagerly)"""


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
        from . import ops
        # (h2 arithmetic: each graph owns the amax words its launches write and read, cleared inside the graph -- ops.h2_private_pool)
        self.h2_vis, self.h2_txt = ops.h2_private_pool(), ops.h2_private_pool()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):  # first-call setup (hipFuncSetAttribute, workspace growth) outside capture
                    v = net.encode_visual(self.s_img)
                    net.forward_cached(v, self.s_ids, H)
            torch.cuda.current_stream().wait_stream(side)
            self.g_vis = torch.cuda.CUDAGraph()
            with self.h2_vis, torch.cuda.graph(self.g_vis):
                self.h2_vis.reset()
                self.vis = net.encode_visual(self.s_img)
            self.g_txt = torch.cuda.CUDAGraph()
            with self.h2_txt, torch.cuda.graph(self.g_txt):
                self.h2_txt.reset()
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
        self.s_ids = torch.zeros(n, L, device=dev, dtype=torch.int64)
        self.s_ids[:, 0] = 49406
        self.s_ids[:, L - 1] = 49407      # (full-length sentences: the warm-up calls -- and the autotuner behind them -- see every row in use)
        from . import ops
        # (h2 products tag their operands with words of a PER-STEP amax pool: a graph that outlives the step must not hold such
        # pointers -- this tower owns its words, cleared inside the graph; its frozen weights have constant words)
        self.h2 = ops.h2_private_pool()
        with torch.no_grad(), self.h2:
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                for _ in range(warmup):
                    clip_model.encode_text_hidden(self.s_ids)
            torch.cuda.current_stream().wait_stream(cap)
            self.g = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (the collective backend's watchdog, the autograd engine of a previous
            # step) may touch the device while this stream is capturing
            with torch.cuda.graph(self.g, capture_error_mode="thread_local"):
                self.h2.reset()
                self.out = clip_model.encode_text_hidden(self.s_ids)

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
        _prime(model, clip_model, optimizer, self.bns, (self.s_img, self.s_ids, self.s_neg), args, priming)
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


def _prime(model, clip_model, optimizer, bns, batch, args, priming, reducer=None):
    """eager forward + backward passes WITHOUT an optimiser step, BatchNorm running statistics put back afterwards: first-
    encounter autotuning, workspaces and allocator pools settle before anything is captured (data parallel: every rank primes,
    SyncBatchNorm exchanges and gradient collectives included -- the gradients are discarded)"""
    from .train_stage1 import _step_body
    keep = [(m.running_mean.clone(), m.running_var.clone(), m._nbt_pending) for m in bns]
    for _ in range(priming):
        _step_body(model, clip_model, optimizer, batch[0], batch[1], batch[2], args, reducer, optimizer_step=False)
    torch.cuda.synchronize()
    with torch.no_grad():
        for m, (rm, rv, nbt) in zip(bns, keep):
            m.running_mean.copy_(rm)
            m.running_var.copy_(rv)
            m._nbt_pending = nbt
    del keep
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _capture(graph, stream, pool, fn):
    """record fn()'s launches on `stream` into `graph` (allocations from `pool`); nothing executes"""
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        graph.capture_begin(pool=pool, capture_error_mode="thread_local")
        try:
            out = fn()
        finally:
            graph.capture_end()
    return out


class _Chain:
    """successive graphs on ONE stream, cut at points chosen while capturing: split() ends the graph being recorded and begins
    the next (the replay loop puts an event record or wait between the two)"""

    def __init__(self, stream, pool):
        self.stream, self.pool, self.graphs = stream, pool, []

    def run(self, fn):
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self._begin()
            try:
                out = fn()
            finally:
                self.graphs[-1].capture_end()
        return out

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self.graphs.append(g)

    def split(self):
        self.graphs[-1].capture_end()
        self._begin()


class SegmentedTrainStep:
    """The Stage-1 training step as a CHAIN of single-stream hipGraphs on the step's three streams.

    Why not one graph (GraphedTrainStep): the ROCm 7.x runtime launches a graph with parallel branches node by node (host time
    as eager) and runs the branches one after the other; only a LINEAR graph takes its packet path (0.65 ms of host time for
    the whole step) -- but a linear step loses the text / weight-gradient overlap (50 vs 45 ms).  So the step is cut where its
    streams fork and join, every piece is captured on ONE stream, and the pieces are replayed on the three streams with
    ordinary events between them -- dependencies only ever cross at graph boundaries, where stream order defines them:

        text    : .. [F_text: TRIS text encoder] ........ [F_aux: frozen aux text tower] ........ [B_text: text encoder backward] (W_late)
        compute : [F_trunk a][F_trunk b] -> wait F_text -> [F_heads + aux ViT] -> wait F_aux -> [losses] -> [B_heads] -> [B_seg n] ..
                  .. -> [B_seg 0] -> [AdamW early] -> join -> [AdamW late]
        wgrad   :                                                                             [W_heads]  [W_seg n]   ...  [W_seg 0]

    (F_text starts behind a trunk stage, TRIS_SEG_TEXT_AT, default layer2 -- the trunk's forward graph is cut there; F_aux
    starts behind the TRIS forward and is joined right before the loss: the issue points of the eager step.)

    The backward is cut at autograd level: ops.cut() (behind the stem and every Bottleneck) hands the next layer a fresh leaf
    during capture, and the capture runs torch.autograd.backward segment by segment, feeding each the .grad of the leaf above.
    A segment's weight-gradient launches are not forked from inside its capture (ops.on_wgrad_stream defers them to this
    object), they become the graph W_seg for the weight-gradient stream, released by an event behind B_seg.  Every tensor a
    deferred launch reads stays referenced until all captures are done, so no later segment's capture can be handed its memory.
    The three streams capture into three private pools (graphs that may run concurrently must not share one).

    Same bookkeeping (static inputs, device-side optimiser scalars, BatchNorm step counters) as GraphedTrainStep; same kernels
    in the same arithmetic as the eager step, so step k of a replayed run equals step k of an eager run bit for bit.

    Data parallel (reducer given): SyncBatchNorm's mailbox exchanges are ordinary single-workgroup kernels whose exchange
    counter lives on the device (csrc/comm.hip) -- they are captured like everything else.  Collectives are NOT: while the
    backward graphs are recorded the reducer's boundary nodes only note behind which graph their segment becomes final
    (release()), and the replay issues those all-reduces -- and the sparse embedding exchange -- eagerly from the reducer's
    stream between two graph replays; finish() + ONE AdamW graph close the step (the early part of AdamW would update
    parameters whose gradients are still on the wire)."""

    def __init__(self, model, clip_model, optimizer, args, example, lr_scheduler=None, priming=2, reducer=None):
        from . import ops
        from .config import cfg
        from .CLIP.clip.model import BatchNorm2d
        from .train_stage1 import stage1_loss_block
        img, ids, neg = example
        self.model, self.clip_model, self.optimizer, self.args, self.sched = model, clip_model, optimizer, args, lr_scheduler
        self.s_img = img.detach().clone()
        self.s_ids = ids.detach().clone()
        self.s_neg = None if neg is None else neg.detach().clone()
        net = model.module if hasattr(model, "module") else model
        self.bns = [m for m in net.modules() if isinstance(m, BatchNorm2d)]
        optimizer.enable_device_hyper()
        self.reducer = reducer if (reducer is not None and reducer.active) else None
        self.released, self.embed_rows = [], None
        _prime(model, clip_model, optimizer, self.bns, (self.s_img, self.s_ids, self.s_neg), args, priming, self.reducer)
        if any(m.process_group is not None for m in self.bns):
            # SyncBatchNorm is capturable through the mailbox transport only (its exchanges are plain kernels); priming has set
            # the transport up -- collectively, so every rank takes the same decision here
            from . import comm
            if any(m is None for m in comm.Mailbox._by_group.values()) or not comm.Mailbox._by_group:
                raise NotCapturable("SyncBatchNorm fell back to torch.distributed collectives: the step runs eagerly")
        nbt = [m._nbt_pending for m in self.bns]
        self.text, self.wg = ops.side_stream("text"), ops._wgrad_stream()
        self.cap = torch.cuda.Stream()           # all compute-stream pieces are captured on this one stream
        pool_c, pool_t, pool_w = (torch.cuda.graph_pool_handle() for _ in range(3))
        G = torch.cuda.CUDAGraph
        self.cuts, self.deferred, self.inline, self.cutting, keep = [], [], False, False, []
        B = self.s_img.shape[0]
        K = self.s_neg.shape[1] if (self.s_neg is not None and args.negative_samples > 0) else 0

        def wgrad_graphs(split=False):
            """the deferred weight-gradient launches of the segment just captured, as a graph for their stream (split: every
            other one goes into a second graph for the text stream, idle by then -- used for the last segments, whose weight
            gradients nothing is left to hide behind) -> (graph | None, graph | None, [arena views written])
            (Round 6: the one-launch segments of the stem alternated between the two streams -- no gain: the step's last 0.7 ms are
            AdamW over everything outside the late span plus the stem's weight gradients, together bound by HBM, not by the queue.)"""
            fns, self.deferred = self.deferred, []
            keep.append(fns)
            parts = [fns[0::2], fns[1::2]] if split and len(fns) > 1 else [fns, []]
            out = []
            self.inline = True
            try:
                for part, stream, pool in zip(parts, (self.wg, self.text), (pool_w, pool_t)):
                    g = None
                    if part:
                        g = G()
                        _capture(g, stream, pool, lambda: [fn() for fn, _, _ in part])
                    out.append(g)
            finally:
                self.inline = False
            return out[0], out[1], [sk for _, _, sk in fns] + [t for _, ts, _ in fns for t in ts if t is not None]

        self.h2_pool = ops.h2_begin_step()   # (h2 arithmetic: the amax pool the captured launches write; zeroed per replay)
        self.h2_aux = ops.h2_private_pool()  # (the frozen aux text tower's own words: its graph clears them itself)
        self.h2_arenas = list(ops._H2.get("arenas", [])) if self.h2_pool is not None else []
        # No garbage collection while streams are capturing: a collection that runs in the middle (any thread -- the autograd engine
        # executes Python too) may destroy hipGraphs / free pool memory of an earlier captured step, which is illegal on a capturing
        # thread and aborts the process.  Collect first, then keep the collector off until the last capture has ended.
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        ops._SEG = self
        try:
            # ---- forward.  Where the two text towers START relative to the trunk is chosen as in the eager step (model_stage1.
            # TRIS.forward, train_stage1.stage1_forward_losses): the TRIS text encoder behind a trunk stage (TRIS_SEG_TEXT_AT,
            # default layer2), the frozen aux tower behind the TRIS forward, joined right before the loss -- each of those
            # points is a cut of the compute stream's forward graph with an event in it.
            def ids_all_():
                ids_all = self.s_ids.long()
                if K > 0:
                    ids_all = torch.cat([ids_all, self.s_neg.long().reshape(B * K, -1)], 0)
                return ids_all
            self.inline = True                   # (the text encoder's weight gradients stay on the text stream)
            self.g_ftext = G()
            hidden = _capture(self.g_ftext, self.text, pool_t, lambda: net.backbone.encode_text(self.s_ids)[1])
            self.inline = False
            self.g_faux = G()
            with torch.no_grad(), self.h2_aux:
                ids_all, f_all = _capture(self.g_faux, self.text, pool_t,
                                          lambda: (self.h2_aux.reset(), (lambda i: (i, clip_model.encode_text_hidden(i)))(ids_all_()))[1])
            at = "layer2"   # the TRIS text encoder starts behind this trunk stage (measured best of stem / layer1..3)
            fwd = _Chain(self.cap, pool_c)
            marks = {}

            def cut_here(name):
                def f():
                    marks[name] = len(fwd.graphs)      # graphs[:n] are behind this point
                    fwd.split()
                return f

            def forward():
                hooks = {at: cut_here("text")} if (at in ("stem", "layer1", "layer2", "layer3") and not net.vit_trunk) else None
                self.cutting = True
                vis = net.encode_visual(self.s_img, hooks)
                self.cutting = False
                cut_here("trunk")()
                h = hidden.detach().requires_grad_()
                cls, _, _, sig_out, _ = net.forward_cached(vis, self.s_ids, self.s_img.shape[2], hidden=h)
                cut_here("heads")()

                def joined():
                    cut_here("loss")()
                    return f_all
                return vis, h, stage1_loss_block(clip_model, self.s_img, cls, sig_out, joined, ids_all, K, args)
            vis, h, self.losses = fwd.run(forward)
            trunk_cuts, self.cuts = self.cuts, []
            self.fwd, self.fwd_marks = fwd.graphs, marks

            # ---- backward, segment by segment (compute stream), each followed by its weight gradients (their stream)
            def b_heads():      # (optimizer.zero_grad() is issued eagerly by the replay, in front of the reducer's order check)
                self.losses[0].backward()
            def take_released():
                keys, self.released = self.released, []
                return keys
            g = G()
            _capture(g, self.cap, pool_c, b_heads)
            self.back = [(g,) + wgrad_graphs()[:2] + (take_released(),)]
            self.inline = True
            self.g_btext = G()
            _capture(self.g_btext, self.text, pool_t, lambda: torch.autograd.backward(hidden, h.grad))
            self.inline = False
            self.text_released = take_released()
            n_late = min(6, len(trunk_cuts))   # segments whose weight gradients are split over the weight-gradient and the text stream
            split = True
            late_sinks = []
            for k, (x, leaf) in enumerate(reversed(trunk_cuts)):
                late = k >= len(trunk_cuts) - n_late
                g = G()
                _capture(g, self.cap, pool_c, lambda: torch.autograd.backward(x, leaf.grad))
                gw, gt, sinks = wgrad_graphs(split=late and split)
                self.back.append((g, gw, gt, take_released()))
                if late:
                    late_sinks += sinks
            self.first_late = len(self.back) - n_late
            # ---- AdamW.  The update is element-wise, so it can be cut where the gradients become final: everything outside the
            # arena span the LATE segments' weight-gradient launches write is complete once the early ones are, and is updated
            # on the compute stream while those last launches still run; the span itself follows the final join.
            spans = self._late_spans(optimizer, late_sinks) if (n_late and self.reducer is None) else None
            self.g_opt_early = None
            if spans:
                early = []
                for gi, a in enumerate(optimizer.arenas):
                    lo, hi = spans.get(gi, (0, 0))
                    early += [(gi, 0, lo), (gi, hi, a.numel)] if hi > lo else [(gi, 0, a.numel)]
                self.g_opt_early = G()
                _capture(self.g_opt_early, self.cap, pool_c, lambda: optimizer.step(device_hyper=True, ranges=early))
                rest = [(gi, lo, hi) for gi, (lo, hi) in spans.items()]
            else:
                rest = None
            # Data parallel: AdamW per reducer segment.  The update is element-wise and a segment's gradients are final the moment
            # its all-reduce has completed, so each segment released DURING the backward is updated right behind its collective,
            # on the reducer's stream, while the backward goes on; only what is reduced in finish() (and the sparsely exchanged
            # embedding table) is left for the AdamW graph behind the join.  (A parameter a later-issued launch still reads -- the
            # gamma / beta a folded BatchNorm's weight-gradient launch re-normalises with -- belongs to a LATER segment, whose
            # collective waits for that launch.  Checked, not only argued: cfg.ddp_seg_poison keeps an updated segment's parameters NaN
            # until the step ends, tests/test_gpu_ddp.py runs the two-rank step that way and finds no NaN.)
            self.g_opt_seg, self.g_opt_ranges, self._poisoned, self.poisoned_segments = {}, {}, [], 0
            red = self.reducer
            if red is not None and cfg.ddp_seg_opt and red.segments:
                released = [k for (_, _, _, keys) in self.back for k in keys] + list(self.text_released)
                pool_o = torch.cuda.graph_pool_handle()
                covered = {}
                for k in released:
                    segs = [(ai, s_, e_) for ai, s_, e_ in red.segments.get(k, []) if e_ > s_]
                    if not segs or k in self.g_opt_seg or sorted(red._dense_ranges(k)) != sorted(segs):
                        continue
                    g = G()
                    _capture(g, self.cap, pool_o, lambda: optimizer.step(device_hyper=True, ranges=segs))
                    self.g_opt_seg[k] = g
                    self.g_opt_ranges[k] = segs
                    for ai, s_, e_ in segs:
                        covered.setdefault(ai, []).append((s_, e_))
                if self.g_opt_seg:
                    rest = []
                    for gi, a in enumerate(optimizer.arenas):
                        cur = 0
                        for s_, e_ in sorted(covered.get(gi, [])):
                            if s_ > cur:
                                rest.append((gi, cur, s_))
                            cur = max(cur, e_)
                        if cur < a.numel:
                            rest.append((gi, cur, a.numel))
            self.g_opt = G()
            _capture(self.g_opt, self.cap, pool_c, lambda: optimizer.step(device_hyper=True, ranges=rest))
        finally:
            ops._SEG = None
            ops.h2_end_step()
            self.cuts, self.deferred = [], []
            if gc_was:
                gc.enable()
        del keep, trunk_cuts, h, hidden, vis, late_sinks, fwd
        for m, n in zip(self.bns, nbt):          # (capture launched nothing: the forward's host-side count is taken back)
            m._nbt_pending = n
        self.ev = [torch.cuda.Event() for _ in range(len(self.back) + 8)]
        self.trace = False       # developer switch: per-piece HIP-event time stamps of the next replay -> marks()
        self._marks = None

    def _reduce_and_update(self, red, key):
        """issue segment `key`'s all-reduce (reducer's stream) and, behind it on the same stream, that segment's AdamW graph"""
        from . import ops
        if key in red.done or not red.active:
            return
        n0 = len(red.pending)
        red._launch_now(key)
        g = self.g_opt_seg.get(key)
        if g is not None:
            with torch.cuda.stream(ops.side_stream("reduce")):
                for h in red.pending[n0:]:
                    h.wait()              # (RCCL: the reducer's stream waits for the backend's, no host synchronisation)
                red.scale_now(key)        # (a backend without an averaging all-reduce: the mean of this segment, now)
                g.replay()
                from .config import cfg
                if cfg.ddp_seg_poison:
                    # Checking mode: this update is only correct if nothing issued from here on reads the segment's
                    # parameters -- the backward goes on while it runs.  The updated values are set aside and the parameters are NaN
                    # until the step's last launch (__call__ puts them back): a later data-gradient or weight-gradient launch that
                    # read one of them would leave NaN in the gradients, which tests/test_gpu_ddp.py looks for.
                    for ai, s_, e_ in self.g_opt_ranges[key]:
                        v = self.optimizer.arenas[ai].p[s_:e_]
                        self._poisoned.append((v, v.clone()))
                        v.fill_(float("nan"))
                    self.poisoned_segments += 1

    @staticmethod
    def _late_spans(optimizer, touched):
        """{group index: (lo, hi)}: per arena, the element range that covers every arena-resident tensor in `touched` -- the
        gradient views the late weight-gradient launches WRITE and the parameters they READ (a BatchNorm folded into a direct
        convolution re-normalises its input with gamma / beta: the early AdamW must not have updated them yet).  Tensors that
        live in no arena (activations) do not constrain anything; None if a gradient view lies in no arena."""
        spans = {}
        for t in touched:
            for gi, a in enumerate(optimizer.arenas):
                for base in (a.g, getattr(a, "p", None)):
                    if base is None:
                        continue
                    off = (t.data_ptr() - base.data_ptr()) // 4
                    if 0 <= off < a.numel and t.device == base.device:
                        lo, hi = spans.get(gi, (off, off))
                        end = (off + t.numel() + 63) // 64 * 64
                        spans[gi] = (min(lo, off // 64 * 64), max(hi, min(end, a.numel)))
                        break
                else:
                    continue
                break
        return spans or None

    def marks(self):
        """[(name, ms since the start of the step)] of the last replay made with .trace = True (synchronises)"""
        torch.cuda.synchronize()
        t0 = self._marks[0][1]
        return [(n, t0.elapsed_time(e)) for n, e in self._marks]

    # -- ops._SEG interface (only while capturing)
    def cut(self, x):
        if not self.cutting:           # (only the trainable trunk is cut: its weight gradients are what the cuts release)
            return x
        leaf = x.detach().requires_grad_()
        for attr in ("_bn_link", "_bn_lazy", "_pl", "_h2"):  # (hand-offs / plane tags that ride on the tensor: the cut is an identity)
            v = getattr(x, attr, None)
            if v is not None:
                setattr(leaf, attr, v)
        self.cuts.append((x, leaf))
        return leaf

    def release(self, key):
        """(GradReducer._launch while capturing) segment `key` is final behind the backward graph being recorded"""
        self.released.append(key)

    def defer(self, fn, tensors, sink):
        if self.inline:
            return fn()
        self.deferred.append((fn, tensors, sink))
        return None

    def __call__(self, img, ids, neg):
        from . import ops
        main, text, wg, ev = torch.cuda.current_stream(), self.text, self.wg, self.ev
        tm = self._marks = [] if self.trace else None

        def mark(name, stream=main):
            if tm is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                tm.append((name, e))
        mark("start")
        if self.h2_pool is not None:    # h2 arithmetic: a fresh amax pool, the weights' amaxes (the captured launches
            self.h2_pool.zero_()        # write / read the same words every replay)
            ops.h2_weights_amax(self.h2_arenas)
        self.s_img.copy_(img, non_blocking=True)
        self.s_ids.copy_(ids, non_blocking=True)
        if self.s_neg is not None:
            self.s_neg.copy_(neg, non_blocking=True)
        opt = self.optimizer
        opt._steps += 1
        opt.push_hyper()
        ev[0].record(main)
        fm = self.fwd_marks
        from .config import cfg as _cfg
        aux_early = bool(_cfg.aux_text_early)
        for i, g in enumerate(self.fwd):
            if i == fm.get("text", 0):          # the TRIS text encoder starts behind this point of the trunk
                if i:
                    ev[0].record(main)
                text.wait_event(ev[0])
                with torch.cuda.stream(text):
                    self.g_ftext.replay()
                    ev[1].record(text)
                    mark("text_fwd_done", text)
                    if aux_early:
                        # the frozen aux text tower depends on the token ids alone: right behind the TRIS text encoder, under the trunk's
                        # large kernels, instead of under the aux ViT -- a chain of small products that is the step's critical path there
                        # and runs faster alone (cfg.aux_text_early; 0: behind the TRIS forward, the eager step's issue point)
                        self.g_faux.replay()
                        ev[7 + len(self.back)].record(text)
                        mark("aux_text_done", text)
            if i == fm["trunk"]:                # trunk issued: the heads need the sentence features
                mark("trunk_fwd_done")
                main.wait_event(ev[1])
            if i == fm["heads"] and not aux_early:   # TRIS forward issued: the frozen aux text tower goes under the aux ViT
                e = ev[6 + len(self.back)]
                e.record(main)
                text.wait_event(e)
                with torch.cuda.stream(text):
                    self.g_faux.replay()
                    ev[7 + len(self.back)].record(text)
            if i == fm["loss"]:
                main.wait_event(ev[7 + len(self.back)])
            g.replay()
        mark("heads_fwd_done")
        wg.wait_event(ev[0])
        red = self.reducer
        opt.zero_grad()
        if red is not None:
            red.begin_step()
        for i, (gb, gw, gt, keys) in enumerate(self.back):
            gb.replay()
            e = ev[6 + i]
            e.record(main)
            mark(f"b{i}_done")
            if i == 0:
                text.wait_event(e)
                with torch.cuda.stream(text):
                    self.g_btext.replay()
                    ev[2].record(text)
                    mark("text_bwd_done", text)
                    if red is not None:      # the text encoder's segments + the sparse embedding exchange: behind its backward
                        if self.embed_rows is not None:
                            red.exchange_rows(*self.embed_rows)
                        for k in self.text_released:
                            self._reduce_and_update(red, k)
            if gw is not None:
                wg.wait_event(e)
                with torch.cuda.stream(wg):
                    gw.replay()
                    mark(f"w{i}_done", wg)
            if gt is not None:
                text.wait_event(e)
                with torch.cuda.stream(text):
                    gt.replay()
                    mark(f"w{i}t_done", text)
            if red is not None:              # segments that became final with this piece of backward (their last weight
                for k in keys:               # gradients were just queued on the side streams: the reducer's stream waits for them)
                    self._reduce_and_update(red, k)
            if i == self.first_late - 1:
                ev[4].record(wg)        # every weight gradient outside the late span is behind this
        if self.g_opt_early is not None:
            main.wait_event(ev[4])
            main.wait_event(ev[2])
            self.g_opt_early.replay()
            mark("opt_early_done")
        ev[3].record(wg)
        ev[5].record(text)
        main.wait_event(ev[3])
        main.wait_event(ev[5])
        mark("joined")
        if red is not None:
            red.finish()                     # what is left (embedding, stem), then the compute stream waits for the collectives
        self.g_opt.replay()
        if self._poisoned:                   # (cfg.ddp_seg_poison; red.finish() has joined the reducer's stream)
            for v, saved in self._poisoned:
                v.copy_(saved)
                saved.record_stream(main)
            self._poisoned = []
        mark("opt_done")
        for m in self.bns:
            m._nbt_pending += 1
        if self.sched is not None:
            self.sched.step()
        return self.losses


def frozen_text(clip_model, ids):
    """encode_text(ids)[1] of a frozen tower through a cached hipGraph (cfg.hipgraph = False: eager)"""
    from . import ops
    from .config import cfg
    if not cfg.hipgraph or torch.cuda.is_current_stream_capturing() or torch.is_grad_enabled():
        return clip_model.encode_text_hidden(ids)
    n, L = ids.shape
    cache = clip_model.__dict__.setdefault("_tris_text_graphs", {})
    key = (n, L, clip_model.token_embedding.weight.data_ptr(), ops.get_gemm_mode(), cfg.text_pack)
    g = cache.get(key)
    if g is None:
        if len(cache) >= 4:
            cache.clear()
        g = cache[key] = GraphedFrozenText(clip_model, n, L)
    return g(ids)
