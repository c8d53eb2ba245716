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
        self._steps = 0

    def zero_grad(self, set_to_none=False):
        # gradients are overwritten (not accumulated) by the backward kernels each step; keep the sinks attached
        return None

    @torch.no_grad()
    def step(self, closure=None):
        self._steps += 1
        ops.wgrad_join()  # weight gradients are produced on their own stream
        st = torch.cuda.current_stream().cuda_stream
        for g, a in zip(self.param_groups, self.arenas):
            b1, b2 = g["betas"]
            _lib.call("tris_adamw_f32", a.p.data_ptr(), a.g.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), a.numel,
                      float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), self._steps, st)

    def state_dict(self):
        sd = {"steps": self._steps, "groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
              "exp_avg": [a.m.clone() for a in self.arenas], "exp_avg_sq": [a.v.clone() for a in self.arenas]}
        return sd

    def load_state_dict(self, sd):
        self._steps = sd["steps"]
        for g, s in zip(self.param_groups, sd["groups"]):
            g.update(s)
        for a, m, v in zip(self.arenas, sd["exp_avg"], sd["exp_avg_sq"]):
            a.m.copy_(m)
            a.v.copy_(v)
