"""Cross-modal attention modules of TRIS on MI355X kernels: `bilateral_prompt` (Stage-1) and `PixelAttention`
(Stage-2's pixel x token attention, reference model/attn.py:9-65).

`bilateral_prompt` -- the image<->text cross-modal attention of TRIS Stage-1 (reference model/attn.py:68-136),
rebuilt on MI355X kernels.  Same constructor, parameter names (`v_proj{1,2,3}.{0,1}`, `t_proj{1,2,3}.0`,
`v_output.{0,1}`, `t_output.0`) and call signature:

    forward(vis [B,C,H,W], lan [B,C,N]) -> (new_vis [B,C,H,W], new_lan [B,N,C])

Internally activations are channels-last ([B,P,C]); `forward_cl` is the copy-free entry TRIS uses.  The sentence set
is identical for every image of the batch in Stage-1 (model_stage1.py:66 repeats it), so `forward_cl` accepts the
un-repeated [N,C] sentence matrix and projects it once instead of B times (same values).
"""
import math

import torch
from torch import nn

from .. import ops
from ..config import cfg
from ..CLIP.clip.model import Conv2d, Linear


class InstanceNorm2d(nn.Module):
    """affine InstanceNorm2d, instance statistics in train and eval (track_running_stats=False)"""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class ReLU(nn.Module):
    pass


def _vbranch(seq, x, relu, grad_box=None, grad_box_out=None):
    """Sequential(Conv2d 1x1 + bias, InstanceNorm2d, [ReLU]) on channels-last [B,P,C]"""
    y = ops.linear(x, seq[0].weight, seq[0].bias, grad_box=grad_box, grad_box_out=grad_box_out)
    return ops.instance_norm(y, seq[1].weight, seq[1].bias, relu, seq[1].eps)


class bilateral_prompt(nn.Module):
    def __init__(self, vis_chans, lan_chans, m_chans=None):
        super().__init__()
        m = vis_chans if m_chans is None else m_chans
        for i in (1, 2, 3):
            setattr(self, f"v_proj{i}", nn.Sequential(Conv2d(vis_chans, m, 1, bias=True), InstanceNorm2d(m), ReLU()))
        for i in (1, 2, 3):
            setattr(self, f"t_proj{i}", nn.Sequential(Linear(lan_chans, m), ReLU()))
        self.v_output = nn.Sequential(Conv2d(m, vis_chans, 1, bias=True), InstanceNorm2d(vis_chans))
        self.t_output = nn.Sequential(Linear(m, lan_chans))

    def forward_cl(self, vis, lan, box_vis=None, box_lan=None):
        """vis [B,P,C] channels-last pixels, lan [N,C] sentences -> (new_vis [B,P,C], new_lan [B,N,C]).
        vis and lan each feed three projections (and, in the caller, a residual mix): instead of autograd summing four gradients
        with element-wise passes, each projection's data-gradient product adds the sum so far in its epilogue (ops.GradBox chain:
        backward runs the projections in reverse creation order; a box that is not filled in time just falls back to autograd).
        box_vis / box_lan: where the caller's later consumer of vis / lan leaves its gradient (or None)."""
        B, Pp, C = vis.shape
        scale = 1.0 / math.sqrt(lan.shape[-1])
        chain = torch.is_grad_enabled() and cfg.grad_box

        def boxes(x, last):
            if not (chain and x.requires_grad):
                return None, None, None
            return ops.GradBox(), ops.GradBox(), last
        v1, v2, v3 = boxes(vis, box_vis)
        t1, t2, t3 = boxes(lan, box_lan)
        Qv = _vbranch(self.v_proj1, vis, True, grad_box=v1)
        Kv = _vbranch(self.v_proj2, vis, True, grad_box=v2, grad_box_out=v1)
        Vv = _vbranch(self.v_proj3, vis, True, grad_box=v3, grad_box_out=v2)
        tp = [getattr(self, f"t_proj{i}")[0] for i in (1, 2, 3)]
        Qt = ops.linear(lan, tp[0].weight, tp[0].bias, None, 1, grad_box=t1)
        Kt = ops.linear(lan, tp[1].weight, tp[1].bias, None, 1, grad_box=t2, grad_box_out=t1)
        Vt = ops.linear(lan, tp[2].weight, tp[2].bias, None, 1, grad_box=t3, grad_box_out=t2)
        if Qt.shape[0] <= 64 and C % 64 == 0 and Pp <= 256:
            new_vis, new_lan = ops.xattn(Qv, Kv, Vv, Qt, Kt, Vt)   # fused cross attention (csrc/xattn.hip)
        else:  # composed from the GEMM core + row softmax (same math) for shapes outside the fused kernel's limits
            Av = ops.softmax(ops.matmul(Qv, Kt, tB=True), scale)        # [B,P,N]  softmax over sentences
            At = ops.softmax(ops.bmm(Qt, Kv, tB=True), scale)           # [B,N,P]  softmax over pixels
            new_vis = ops.matmul(Av, Vt, tB=False)                      # [B,P,C]
            new_lan = ops.bmm(At, Vv, tB=False)                         # [B,N,C]
        new_vis = _vbranch(self.v_output, new_vis, False)
        new_lan = self.t_output[0](new_lan)
        return new_vis, new_lan

    def forward_pairs(self, vis, lan, owner):
        """Evaluation of (image, sentence) PAIRS in one pass (tris_amd.validate): vis [G,P,C] image features, lan [S,C]
        sentences, owner[k] = image of sentence k.  Every pair is attended exactly as the reference's per-sentence call does
        (a sentence SET of one: validate.py:173-179 runs the model per sentence), so the soft-max over sentences sees one
        sentence -- the projections and the output branches are batched over images / pairs, the cross attention itself is
        launched per pair on views of the batched projections.  -> (new_vis [S,P,C], new_lan [S,1,C])"""
        Qv, Kv, Vv = (_vbranch(getattr(self, f"v_proj{i}"), vis, True) for i in (1, 2, 3))
        Qt, Kt, Vt = (getattr(self, f"t_proj{i}")[0](lan, act=1) for i in (1, 2, 3))
        nv, nl = [], []
        for k, i in enumerate(owner):
            a, b = ops.xattn(Qv[i:i + 1], Kv[i:i + 1], Vv[i:i + 1], Qt[k:k + 1], Kt[k:k + 1], Vt[k:k + 1])
            nv.append(a)
            nl.append(b)
        new_vis = _vbranch(self.v_output, torch.cat(nv, 0), False)
        new_lan = self.t_output[0](torch.cat(nl, 0))
        return new_vis, new_lan

    def forward_sets(self, vis, lan):
        """vis [B,P,C], lan [B,N,C]: EVERY image attends over its OWN sentence set -- what the reference module computes for any
        `lan [B,C,N]` (model/attn.py:111-136); Stage-1 itself only ever passes one set repeated (forward_cl).  Composed from the
        batched products of the GEMM core and the row soft-max: image b's logits are Qv[b] . Kt[b]^T / sqrt(C) (soft-max over its
        N sentences) and Qt[b] . Kv[b]^T / sqrt(C) (soft-max over its P pixels)."""
        B, Pp, C = vis.shape
        N = lan.shape[1]
        scale = 1.0 / math.sqrt(lan.shape[-1])
        Qv, Kv, Vv = (_vbranch(getattr(self, f"v_proj{i}"), vis, True) for i in (1, 2, 3))
        flat = lan.reshape(B * N, lan.shape[-1])
        Qt, Kt, Vt = (getattr(self, f"t_proj{i}")[0](flat, act=1).reshape(B, N, -1) for i in (1, 2, 3))
        Av = ops.softmax(ops.bmm(Qv, Kt, tB=True), scale)           # [B,P,N]  soft-max over the image's sentences
        At = ops.softmax(ops.bmm(Qt, Kv, tB=True), scale)           # [B,N,P]  soft-max over the image's pixels
        new_vis = ops.bmm(Av, Vt, tB=False)                         # [B,P,C]
        new_lan = ops.bmm(At, Vv, tB=False)                         # [B,N,C]
        new_vis = _vbranch(self.v_output, new_vis, False)
        new_lan = self.t_output[0](new_lan)
        return new_vis, new_lan

    def forward(self, vis, lan):
        B, C, H, W = vis.shape
        lan_t = lan.transpose(1, 2)  # [B,N,C]
        vis_cl = vis.permute(0, 2, 3, 1).reshape(B, H * W, C)
        if B > 1 and not bool((lan_t[0:1] == lan_t).all()):
            nv, nl = self.forward_sets(vis_cl.contiguous(), lan_t.contiguous())   # per-image sentence sets: the general form
        else:
            nv, nl = self.forward_cl(vis_cl, lan_t[0].contiguous())               # one set shared by all images (Stage-1)
        return nv.reshape(B, H, W, C).permute(0, 3, 1, 2), nl


class Conv1d(nn.Module):
    """kernel-size-1 Conv1d (a per-token Linear) keeping the reference's [Cout, Cin, 1] weight shape"""

    def __init__(self, cin, cout):
        super().__init__()
        w = torch.empty(cout, cin, 1)
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x, act=0):  # x [..., Cin] token-major
        return ops.linear(x, self.weight, self.bias, act=act)


class PixelAttention(nn.Module):
    """Stage-2 pixel x token attention (model/attn.py:9-65): every pixel attends over the T word features of its own
    sentence, the attended language feature gates the visual feature.  Same constructor, parameter names
    (Wk, Wv, Wq, Wm, Ww, Wo, ins_q, ins_w) and call signature:

        forward(vis_feat [N,Ci,H,W], lan_feat [N,Ct,T]) -> [N,Ci,H,W]

    Built from the Stage-1 kernel family: the six 1x1 projections are MFMA GEMMs with fused bias / ReLU, the two
    [P,T] / [T,Ci] products run as batched GEMMs, softmax / InstanceNorm / gating are the Stage-1 kernels."""

    def __init__(self, visual_channel, language_channel):
        super().__init__()
        self.Ci, self.Ct = visual_channel, language_channel
        self.Wk = Conv1d(self.Ct, self.Ci)
        self.Wv = Conv1d(self.Ct, self.Ci)
        self.Wq = Conv2d(self.Ci, self.Ci, 1, bias=True)
        self.Wm = Conv2d(self.Ci, self.Ci, 1, bias=True)
        self.Ww = Conv2d(self.Ci, self.Ci, 1, bias=True)
        self.Wo = Conv2d(self.Ci, self.Ci, 1, bias=True)
        self.ins_q = InstanceNorm2d(self.Ci)
        self.ins_w = InstanceNorm2d(self.Ci)

    def forward_cl(self, x, lan):
        """x [N,P,Ci] channels-last pixels, lan [N,T,Ct] token-major words -> [N,P,Ci]"""
        Lk, Lv = self.Wk(lan), self.Wv(lan)                                             # [N,T,Ci]
        Vq = ops.instance_norm(ops.linear(x, self.Wq.weight, self.Wq.bias), self.ins_q.weight, self.ins_q.bias, False,
                               self.ins_q.eps)
        attn = ops.softmax(ops.bmm(Vq, Lk, tB=True), 1.0 / math.sqrt(self.Ci))          # [N,P,T] over the words
        G = ops.bmm(attn, Lv, tB=False)                                                 # [N,P,Ci]
        Gi = ops.instance_norm(ops.linear(G, self.Ww.weight, self.Ww.bias), self.ins_w.weight, self.ins_w.bias, False,
                               self.ins_w.eps)
        Vo = ops.linear(x, self.Wm.weight, self.Wm.bias, act=1)
        return ops.linear(ops.mul(Vo, Gi), self.Wo.weight, self.Wo.bias, act=1)

    def forward(self, vis_feat, lan_feat):
        N, Ci, H, W = vis_feat.shape
        out = self.forward_cl(vis_feat.permute(0, 2, 3, 1).reshape(N, H * W, Ci), lan_feat.transpose(1, 2).contiguous())
        return out.reshape(N, H, W, Ci).permute(0, 3, 1, 2)
