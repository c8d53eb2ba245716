"""TRIS Stage-1 model on MI355X kernels -- drop-in for the reference's `model.model_stage1.TRIS`
(/root/reference/model/model_stage1.py:14-123): same constructor (`TRIS(args)`), the same 518 state-dict keys,
`.backbone.encode_text/encode_image`, `.trainable_parameters()`, and

    forward(x [B,3,H,W], word_id [B,L]) -> train: (cls_out [B,B], cls_fg [B], relu_map, sigmoid_map, exp(logit_scale))
                                           eval : relu_map [B,1,H,W]
"""
import numpy as np
import torch
from torch import nn

from .. import ops
from ..config import cfg
from ..CLIP import clip
from ..CLIP.clip.model import Conv2d, Linear
from .attn import bilateral_prompt


_SIDE = {}


def _overlap_enabled():
    return cfg.text_stream and ops.streams_allowed()


def _side_stream(device):
    return ops.side_stream("text")


def _text_issue_point():
    """start | stem | layer1 | layer2 | layer3 | layer4: after which trunk stage the text encoder is issued (TRIS_TEXT_AT)"""
    at = cfg.text_at
    return at if at in ("start", "stem", "layer1", "layer2", "layer3", "layer4") else "layer4"


class TRIS(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.bert_model = args.bert_tokenizer
        if args.backbone == "clip-RN50":
            last_vis_channel, self.textdim = 2048, 1024
        elif args.backbone == "clip-RN101":
            last_vis_channel, self.textdim = 2048, 512
        elif args.backbone in ("clip-ViT-B/16", "clip-ViT-B/32"):
            # BASELINE config 5.  NOT defined by the reference (model_stage1.py:20-25 leaves last_vis_channel unbound for
            # it): the definition here follows the variant sketched in comments at CLIP/clip/model.py:427-441 -- spatial
            # tokens after the last block, interpolated positional embedding -- with the text width of the ViT-B CLIPs.
            last_vis_channel, self.textdim = 768, 512
        else:
            raise ValueError(f"backbone {args.backbone!r} has no Stage-1 definition in the reference "
                             "(model_stage1.py:20-25 covers clip-RN50 / clip-RN101 only)")
        self.vit_trunk = "ViT" in args.backbone
        device = "cuda" if torch.cuda.is_available() else "cpu"
        clip_model, _ = clip.load(args.backbone.split("-", 1)[-1] if self.vit_trunk else args.backbone.split("-")[-1],
                                  device=device, jit=False, txt_length=args.max_query_len)
        self.backbone = clip_model.float()
        if self.vit_trunk:
            # the dense trunk stops after the last block (forward_spatial): ln_post / proj never receive a gradient, so --
            # like torch.optim.AdamW, which skips parameters whose grad is None -- they must stay out of the optimiser
            # arenas (no weight decay on them either)
            for p in list(self.backbone.visual.ln_post.parameters()) + [self.backbone.visual.proj]:
                p._tris_no_grad_path = True
        self.vis_project = Conv2d(last_vis_channel, args.hidden_dim, 1, bias=True)
        self.lan_project = Linear(self.textdim, args.hidden_dim)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        if self.args.attn_multi > 0:
            self.attn_fusion = bilateral_prompt(args.hidden_dim, lan_chans=args.hidden_dim)

    def trainable_parameters(self):
        new = [self.vis_project, self.lan_project]
        if hasattr(self, "attn_fusion"):
            new.append(self.attn_fusion)
        return list(self.backbone.parameters()), list(nn.ModuleList(new).parameters())

    def encode_visual(self, x, hooks=None):
        """Image-only half of forward (RN50 trunk -> vis_project -> L2 norm).  Returned state can be reused for every
        sentence of the same image (validate.py re-runs the trunk per sentence; the values are identical)."""
        B = x.shape[0]
        if self.vit_trunk:
            if hooks:
                for h in hooks.values():
                    h()
            c4 = self.backbone.visual.forward_spatial(x)[1]             # [B,h,w,768] channels-last
        else:
            c4 = self.backbone.visual.forward_cl(x, hooks, taps=False)[3]   # [B,h,w,2048] channels-last
        h_, w_ = c4.shape[1:3]
        vis = self.vis_project(c4).reshape(B, h_ * w_, -1)              # [B,P,C]
        return ops.l2norm(vis), h_, w_

    def forward_cached(self, vis_state, word_id, out_size, hidden=None):
        norm_vis, h_, w_ = vis_state
        B = norm_vis.shape[0]
        if hidden is None:
            _, hidden = self.backbone.encode_text(word_id)             # [N,E]   (N = B sentences)
        norm_lan = ops.l2norm(self.lan_project(hidden))                 # [N,C]
        if self.args.attn_multi > 0:
            # norm_vis / norm_lan feed the three projections of the fusion AND the residual mix below: the mix leaves its gradient
            # in a box that the fusion's last projection adds in its data-gradient epilogue (attn.forward_cl)
            grad = torch.is_grad_enabled() and cfg.grad_box
            bv = ops.GradBox() if grad and norm_vis.requires_grad else None
            bl = ops.GradBox() if grad and norm_lan.requires_grad else None
            new_vis, new_lan = self.attn_fusion.forward_cl(norm_vis, norm_lan, box_vis=bv, box_lan=bl)
            norm_vis = ops.axpy(new_vis, norm_vis, 0.1, grad_box_b=bv)  # hard-coded 0.1 (model_stage1.py:73-74)
            lan_b = ops.axpy_bcast(new_lan, norm_lan, 0.1, grad_box_b=bl)   # [B,N,C]: 0.1 * new_lan + norm_lan per image
        else:
            lan_b = norm_lan.unsqueeze(0).expand(B, -1, -1).contiguous()
        score = ops.bmm(norm_vis, lan_b, tB=True)                       # [B,P,N]
        score, logit_scale = ops.scale_exp(score, self.logit_scale)    # score * logit_scale.exp(), and the scale itself
        if self.training:
            cls_out, cls_fg, relu_map, sig_map = ops.score_heads(score, h_, w_, out_size, True,
                                                                 float(self.args.FOCAL_P),
                                                                 float(self.args.FOCAL_LAMBDA))
            return cls_out, cls_fg, relu_map, sig_map, logit_scale
        return ops.score_heads(score, h_, w_, out_size, False)

    @torch.no_grad()
    def forward_pairs(self, vis_state, word_id, owner, out_size):
        """Evaluation: response maps of S (image, sentence) pairs in one pass.  vis_state = encode_visual of G images,
        word_id [S, L], owner[k] = index of sentence k's image.  Each pair is computed as the reference computes it -- the
        model called with ONE sentence (validate.py:173-179) -- so under ops.batch_invariant() map k equals
        forward_cached(encode_visual(img[owner[k]]), word_id[k:k+1]) bit for bit.  -> relu maps [S,1,H,W]"""
        assert not self.training
        norm_vis, h_, w_ = vis_state
        S = word_id.shape[0]
        idx = torch.as_tensor(list(owner), device=norm_vis.device, dtype=torch.long)
        _, hidden = self.backbone.encode_text(word_id)                 # [S,E]
        norm_lan = ops.l2norm(self.lan_project(hidden))                 # [S,C]
        vis_p = norm_vis.index_select(0, idx)                           # [S,P,C]  (a copy of each pair's image features)
        if self.args.attn_multi > 0:
            new_vis, new_lan = self.attn_fusion.forward_pairs(norm_vis, norm_lan, list(owner))
            vis_p = ops.axpy(new_vis, vis_p, 0.1)
            lan_p = ops.axpy(new_lan, norm_lan.unsqueeze(1).contiguous(), 0.1)      # [S,1,C]
        else:
            lan_p = norm_lan.unsqueeze(1).contiguous()
        score = ops.scale_exp(ops.bmm(vis_p, lan_p, tB=True), self.logit_scale)[0]   # [S,P,1]
        # the map kernels read score[i,:,i] (Stage-1's diagonal): lay the S single-sentence columns out on a diagonal
        full = torch.zeros(S, score.shape[1], S, device=score.device, dtype=torch.float32)
        ar = torch.arange(S, device=score.device)
        full[ar, :, ar] = score[:, :, 0]
        return ops.score_heads(full, h_, w_, out_size, False)

    def forward(self, x, word_id):
        if self.training:
            ops.h2_auto_step()     # (h2 operand planes: a forward outside train_step's bracket runs under an amax pool of its own)
        with ops.h2_auto_lock():
            return self._forward(x, word_id)

    def _forward(self, x, word_id):
        if not _overlap_enabled():
            return self.forward_cached(self.encode_visual(x), word_id, x.shape[2])
        # The text encoder (short GEMMs that cannot fill 256 CUs) runs on a second HIP stream, concurrently with the
        # RN50 trunk; autograd replays each branch's backward on the stream its forward ran on, so the overlap holds
        # in both directions.  Joined before the heads.
        # WHERE in the issue order the text encoder goes decides where its backward goes: autograd replays nodes in reverse
        # creation order, so a text encoder issued first has its backward issued last -- after the whole trunk backward, as
        # a tail of small kernels with nothing left to overlap (and its gradients are the last to reach the data-parallel
        # reducer).  Issued behind layer<k> of the trunk instead (default: behind layer4, i.e. after the whole trunk has been
        # ISSUED -- the host runs several ms ahead of the GPU, so the text forward still executes under the trunk's tail), its
        # backward is issued before / in the middle of the trunk backward and executes under the BatchNorm-heavy layer1/2/stem
        # part.  Measured (B = 48, same box, img/s): start 968-970 | stem 970 | layer1 987 | layer2 996-1000 | layer3 1003;
        # second box: layer2 1024-1027 | layer3 1025-1027 | layer4 1031.
        main = torch.cuda.current_stream()
        side = _side_stream(x.device)
        box = {}
        ready = torch.cuda.Event()
        ready.record(main)             # inputs and parameters are complete here: the text encoder depends on nothing later

        def issue_text():
            side.wait_event(ready)
            with torch.cuda.stream(side):
                box["hidden"] = self.backbone.encode_text(word_id)[1]
        at = _text_issue_point()
        if at == "start":
            issue_text()
            vis = self.encode_visual(x)
        else:
            vis = self.encode_visual(x, hooks={at: issue_text})
        hidden = box["hidden"]
        main.wait_stream(side)
        hidden.record_stream(main)
        return self.forward_cached(vis, word_id, x.shape[2], hidden=hidden)


def focal_loss(x, p=1, c=0.1):
    return torch.pow(1 - x, p) * torch.log(c + x)
