"""Flat parameter arenas + fused AdamW for the Stage-1 step (reference: torch.optim.AdamW over two parameter
groups, train_stage1.py:135-139, stepped at :370; LambdaLR poly schedule :141-144).

MI355X-first layout: every parameter group lives in four contiguous fp32 arenas (param / grad / exp_avg /
exp_avg_sq).  Parameters become views into the param arena (conv weights keep channels_last strides), their
`.grad` is a view into the grad arena and is written *directly* by the backward kernels (tris_amd.ops "sinks").
One kernel launch updates a whole group, and the grad arena is what data-parallel training all-reduces
(tris_amd.parallel) -- a handful of large RCCL collectives instead of ~350 small ones.

Parameters flagged `_tris_no_grad_path` (attnpool, backbone.logit_scale: never reached by the Stage-1 loss) are
left out, matching torch.optim.AdamW, which skips parameters whose grad is None (no weight decay either).
"""
import torch

from . import _lib, ops

ALIGN = 64  # floats (256 B)


class Arena:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad and not getattr(p, "_tris_no_grad_path", False)]
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = n
        self.p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets = offs
        for p, o in zip(self.params, offs):
            pv, gv = self._view(self.p, p, o), self._view(self.g, p, o)
            with torch.no_grad():
                pv.copy_(p.data)
            p.data = pv
            p.grad = gv
            p._tris_sink = True
            # Backward kernels OVERWRITE the arena gradient of a parameter they are handed directly.  A parameter that
            # reaches its kernels through torch ops instead (the ViT trunk's class / positional embedding: slicing,
            # interpolation, concatenation) gets its gradient from autograd's AccumulateGrad, which ADDS into the arena
            # view: the model flags those (`_tris_accumulates`) and zero_grad() clears them.  (A post-accumulate hook
            # cannot discover them at run time: torch calls it for sunk parameters too.)
            p._tris_accumulates = bool(getattr(p, "_tris_accumulates", False))

    @staticmethod
    def _view(flat, p, off):
        seg = flat[off:off + p.numel()]
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            co, ci, kh, kw = p.shape
            return seg.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return seg.view(p.shape)


class FusedAdamW(torch.optim.Optimizer):
    """AdamW(params_or_groups, lr, betas, eps, weight_decay) with torch semantics, one launch per group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.arenas = []
        for g in self.param_groups:
            if not all(p.is_cuda for p in g["params"]):
                raise ops.NoGpuError("FusedAdamW needs the model on the GPU before construction")
            self.arenas.append(Arena(g["params"]))
        for a in self.arenas:   # (TRIS_LINEAR_MODE=h2: the weights' amaxes in one launch per arena and step)
            ops.h2_register_arena(a)
        self._steps = 0
        self._hyper = None      # device copy of the step-dependent scalars (enable_device_hyper)

    # ---- step-dependent scalars in device memory (captured training step, tris_amd.graphs.GraphedTrainStep) ----------
    # A hipGraph replays its launches with the kernel arguments frozen at capture time, so lr and the two bias corrections
    # cannot be host scalars there: they live in a small device tensor that the host refreshes before every replay through
    # a ring of pinned staging rows (the host may run many steps ahead of the device; a row is reused only after the copy
    # that read it has completed).
    RING = 64

    def enable_device_hyper(self):
        if self._hyper is None:
            dev = self.arenas[0].p.device
            n = len(self.param_groups)
            self._hyper = torch.zeros(n, 4, device=dev, dtype=torch.float32)
            self._ring = torch.zeros(self.RING, n, 4, dtype=torch.float32).pin_memory()
            self._ring_ev = [None] * self.RING
            self._ring_i = 0
        return self._hyper

    def push_hyper(self, step_count=None):
        """{lr, 1 - beta1^t, sqrt(1 - beta2^t)} of every group for the 1-based step t -> device (async, current stream)"""
        import math
        import struct
        t = self._steps if step_count is None else step_count
        slot = self._ring_i % self.RING
        self._ring_i += 1
        ev = self._ring_ev[slot]
        if ev is not None:
            ev.synchronize()
        row = self._ring[slot]
        for gi, g in enumerate(self.param_groups):
            # (the betas as the kernels receive them -- C floats -- so that these are the very numbers tris_adamw_f32 computes)
            b1, b2 = (struct.unpack("f", struct.pack("f", b))[0] for b in g["betas"])
            row[gi, 0] = float(g["lr"])
            row[gi, 1] = 1.0 - b1 ** t
            row[gi, 2] = math.sqrt(1.0 - b2 ** t)
        self._hyper.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ring_ev[slot] = ev

    def zero_grad(self, set_to_none=False):
        """Gradients written by the backward kernels are overwritten each step (the sinks stay attached, nothing to
        clear); the few parameters whose gradient arrives through autograd's AccumulateGrad (seen at run time:
        `_tris_accumulates`) are zeroed here so that they do not sum over steps."""
        for a in self.arenas:
            for p in a.params:
                if p._tris_accumulates:
                    p.grad.zero_()
        return None

    def span(self, params):
        """(group index, lo, hi): the arena range [lo, hi) that holds exactly `params`, or None if they are not one contiguous
        run of one group's arena (used to update part of a group early, tris_amd.graphs.SegmentedTrainStep)"""
        want = {id(p) for p in params if getattr(p, "_tris_sink", False)}
        for gi, a in enumerate(self.arenas):
            idx = [i for i, p in enumerate(a.params) if id(p) in want]
            if not idx:
                continue
            if len(idx) != len(want) or idx != list(range(idx[0], idx[-1] + 1)):
                return None
            hi = a.offsets[idx[-1] + 1] if idx[-1] + 1 < len(a.params) else a.numel
            return gi, a.offsets[idx[0]], hi
        return None

    @torch.no_grad()
    def step(self, closure=None, device_hyper=False, ranges=None):
        """device_hyper=True: the launches read lr / bias corrections from the device tensor of enable_device_hyper() and the
        step counter is NOT advanced here (the caller -- a captured step's replay loop -- advances it and calls push_hyper).
        ranges (device_hyper only): [(group index, lo, hi)] arena ranges to update instead of every group in full (the update is
        element-wise: any partition of the arenas gives the same result as one launch per group)."""
        ops.wgrad_join()  # weight gradients are produced on their own stream
        st = torch.cuda.current_stream().cuda_stream
        if device_hyper:
            hy = self.enable_device_hyper()
            if ranges is None:
                ranges = [(gi, 0, a.numel) for gi, a in enumerate(self.arenas)]
            for gi, lo, hi in ranges:
                if hi <= lo:
                    continue
                g, a = self.param_groups[gi], self.arenas[gi]
                b1, b2 = g["betas"]
                o = lo * 4
                _lib.call("tris_adamw_dev_f32", a.p.data_ptr() + o, a.g.data_ptr() + o, a.m.data_ptr() + o, a.v.data_ptr() + o,
                          hi - lo, hy[gi].data_ptr(), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), st)
            return
        assert ranges is None
        self._steps += 1
        for g, a in zip(self.param_groups, self.arenas):
            b1, b2 = g["betas"]
            _lib.call("tris_adamw_f32", a.p.data_ptr(), a.g.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), a.numel,
                      float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), self._steps, st)

    # ---- checkpoint format: torch.optim.AdamW's own layout -----------------------------------------------------------
    # The reference saves `optimizer.state_dict()` of torch.optim.AdamW under 'optimizer' (utils/util.py:50-64) and loads
    # it back on --resume (:83-95).  state_dict() therefore emits exactly that layout -- {'state': {param index:
    # {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [indices]}]} with indices running over the
    # groups' parameters in order -- and load_state_dict() accepts it (from either code base) by scattering the
    # per-parameter moments into the arenas.  Parameters outside the arenas (no gradient path) have no state entry, as in
    # torch, which creates state lazily for parameters that received a gradient.
    def _index(self):
        """[(group idx, arena, {id(p): offset}), first param index]"""
        out, base = [], 0
        for gi, (g, a) in enumerate(zip(self.param_groups, self.arenas)):
            out.append((gi, g, a, {id(p): o for p, o in zip(a.params, a.offsets)}, base))
            base += len(g["params"])
        return out

    def state_dict(self):
        state, groups = {}, []
        for gi, g, a, offs, base in self._index():
            if self._steps > 0:
                for j, p in enumerate(g["params"]):
                    o = offs.get(id(p))
                    if o is None:
                        continue
                    state[base + j] = {"step": torch.tensor(float(self._steps)),
                                       "exp_avg": Arena._view(a.m, p, o).detach().clone(),
                                       "exp_avg_sq": Arena._view(a.v, p, o).detach().clone()}
            pg = {k: v for k, v in g.items() if k != "params"}
            pg["params"] = list(range(base, base + len(g["params"])))
            groups.append(pg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:   # round-1 layout of this code base (flat arenas)
            self._steps = sd["steps"]
            for g, s in zip(self.param_groups, sd["groups"]):
                g.update(s)
            for a, m, v in zip(self.arenas, sd["exp_avg"], sd["exp_avg_sq"]):
                a.m.copy_(m)
                a.v.copy_(v)
            return
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError(f"optimizer state has {len(sd['param_groups'])} parameter groups, this optimizer "
                             f"{len(self.param_groups)}")
        steps = set()
        for (gi, g, a, offs, base), sg in zip(self._index(), sd["param_groups"]):
            if len(sg["params"]) != len(g["params"]):
                raise ValueError(f"parameter group {gi}: {len(sg['params'])} parameters in the checkpoint, "
                                 f"{len(g['params'])} in the model")
            g.update({k: v for k, v in sg.items() if k != "params"})
            a.m.zero_()
            a.v.zero_()
            for j, (p, idx) in enumerate(zip(g["params"], sg["params"])):
                st = sd["state"].get(idx)
                o = offs.get(id(p))
                if st is None or o is None:
                    continue
                with torch.no_grad():
                    Arena._view(a.m, p, o).copy_(st["exp_avg"].reshape(p.shape))
                    Arena._view(a.v, p, o).copy_(st["exp_avg_sq"].reshape(p.shape))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            # one fused launch uses one bias-correction step count; torch keeps it per parameter.  They only differ when
            # parameters joined training at different times, which the Stage-1 recipe never does.
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not representable by the fused AdamW")
        self._steps = steps.pop() if steps else 0
