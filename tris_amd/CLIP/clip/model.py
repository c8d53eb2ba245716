"""CLIP encoders of the TRIS Stage-1 path on MI355X kernels.

Mirrors the module tree, constructor signatures and the state-dict keys of the reference's modified
CLIP (/root/reference/CLIP/clip/model.py) so checkpoints load unchanged, but every forward is built
from `tris_amd.ops` (HIP kernels, channels-last activations).  Nothing here runs on CPU: modules can
be constructed, moved and (de)serialised anywhere, `forward` needs the GPU.

  Bottleneck              CLIP/clip/model.py:10-55
  AttentionPool2d         :58-104   (parameters kept for checkpoints; output is discarded by Stage-1)
  ModifiedResNet          :195-279  (returns (c1,c2,c3,c4,[global,local]))
  ResidualAttentionBlock  :366-386,  Transformer :389-397,  VisionTransformer :400-448
  CLIP                    :451-580  (encode_text returns (all tokens, projected EOT token) :552-564)
  build_model             :607-644
"""
from collections import OrderedDict


import numpy as np
import torch
from torch import nn

from ... import ops
from ...config import cfg


class Conv2d(nn.Module):
    """Bias-free conv holding its weight as [Cout,Cin,k,k] in channels_last memory (kernel layout)."""

    def __init__(self, cin, cout, k, stride=1, bias=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        w = torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last)
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def forward(self, x, stats=False, grad_box=None, grad_box_out=None):
        """x channels-last [B,H,W,Cin]; stats: fuse the following BatchNorm's statistics; grad_box / grad_box_out: the
        block's ops.GradBox (consume a deposited gradient in the dgrad epilogue / deposit this layer's input gradient)"""
        if self.k == 1:
            return ops.linear(x, self.weight, self.bias, stats=stats and self.bias is None, grad_box=grad_box,
                              grad_box_out=grad_box_out)
        assert self.k == 3 and self.bias is None
        return ops.conv3x3(x, self.weight, self.stride, stats=stats)


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d-compatible parameters/buffers; `process_group` set => SyncBatchNorm semantics."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.process_group = None
        # The step counter is bookkeeping only (momentum is fixed): a 1-element device add per layer per step is 55 extra
        # launches on the critical stream, so forward counts on the host and the buffer is brought up to date whenever
        # it can be observed (state_dict / checkpoint) or replaced (load_state_dict).
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(lambda m, prefix, keep_vars: m.flush_batches_tracked())
        self.register_load_state_dict_post_hook(lambda m, incompatible: setattr(m, "_nbt_pending", 0))

    def flush_batches_tracked(self):
        if self._nbt_pending:
            self.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    def forward(self, x, resid=None, relu=False, grad_box=None, lazy=False, bwd_link=False, pool=False, planes=False, dx_planes=False):
        """pool=True: returns avgpool2(relu(bn(x))) -- the AvgPool2d(2) that follows is part of the op (train mode: one kernel,
        the full-size activation is never written).
        bwd_link=True: the caller guarantees that the output has exactly ONE autograd consumer; if that is a 1x1 convolution
        its data gradient does this BatchNorm's backward reduction in its epilogue (ops._BnBwdLink)"""
        if self.training:
            self._nbt_pending += 1
        group = self.process_group if self.training else None
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, resid, relu,
                              self.training, self.momentum, self.eps, group, grad_box, lazy, bwd_link, pool, planes, dx_planes)


class AvgPool2d(nn.Module):
    def __init__(self, k):
        super().__init__()
        assert k == 2
        self.k = k

    def forward(self, x, grad_box_out=None):
        return ops.avgpool2(x, grad_box_out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3)
        self.bn2 = BatchNorm2d(planes)
        self.avgpool = AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = Conv2d(planes, planes * self.expansion, 1)
        self.bn3 = BatchNorm2d(planes * self.expansion)
        self.downsample = None
        self.stride = stride
        if stride > 1 or inplanes != planes * Bottleneck.expansion:
            self.downsample = nn.Sequential(OrderedDict([
                ("-1", AvgPool2d(stride) if stride > 1 else nn.Identity()),
                ("0", Conv2d(inplanes, planes * self.expansion, 1)),
                ("1", BatchNorm2d(planes * self.expansion)),
            ]))

    def forward(self, x, link_out=False):  # channels-last
        """link_out: the block's output goes to the next block (or one 1x1 convolution) and nowhere else -- its identity use
        there rides a GradBox -- so bn3's backward reduction may be fused into that consumer's data gradient (needs the
        GradBox path: with TRIS_GRAD_BOX=0 the residual gradient reaches the output through autograd as a second term)"""
        tr = self.training  # train mode: BatchNorm batch statistics come out of the producing conv's epilogue
        link_out = link_out and cfg.grad_box
        # identity block: x feeds conv1 and the residual add; the residual gradient rides conv1's data-gradient epilogue
        # down-sampling block: the shortcut's input gradient (avg-pool or 1x1 conv backward) rides along the same way
        box = ops.GradBox() if (tr and torch.is_grad_enabled() and x.requires_grad
                               and cfg.grad_box) else None
        out = self.conv1(x, stats=tr, grad_box=box)
        # bn1 + ReLU feed conv2 only: where direct kernels serve conv2 (forward and weight gradient) the normalised tensor is
        # never written -- they normalise conv1's raw output while staging it (ops.batch_norm lazy=True)
        # (h2 with operand planes: every BatchNorm of the block writes its output -- and the gradient of its input, which only the
        #  backward of the convolution before it reads -- as fp16 piece planes, ops.batch_norm planes / dx_planes; the lazy form is off then)
        pl = tr and ops.planes_on()
        out = self.bn1(out, relu=True, lazy=tr and not pl and ops.conv3x3_bnin_ok(out.shape, self.conv2.cout), bwd_link=True,
                       planes=pl, dx_planes=pl)
        if self.stride > 1:   # bn2 + ReLU + AvgPool2d(stride) as one op
            out = self.bn2(self.conv2(out, stats=tr), relu=True, pool=True, planes=pl, dx_planes=pl)
        else:
            out = self.bn2(self.conv2(out, stats=tr), relu=True, bwd_link=True, planes=pl, dx_planes=pl)   # one consumer: conv3
        out = self.conv3(out, stats=tr)
        if self.downsample is not None:
            if isinstance(self.downsample[0], AvgPool2d):
                idn = self.downsample[1](self.downsample[0](x, grad_box_out=box), stats=tr)
            else:
                idn = self.downsample[1](self.downsample[0](x), stats=tr, grad_box_out=box)
            idn = self.downsample[2](idn, dx_planes=pl)     # (the shortcut's output stays fp32: bn3's pass is its only reader)
            return self.bn3(out, resid=idn, relu=True, bwd_link=link_out, planes=pl, dx_planes=pl)
        return self.bn3(out, resid=x, relu=True, grad_box=box, bwd_link=link_out, planes=pl, dx_planes=pl)  # relu(bn3(conv3) + identity), fused


class Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.in_features, self.out_features = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x, resid=None, act=0, grad_box_res=None):
        return ops.linear(x, self.weight, self.bias, resid, act, grad_box_res=grad_box_res)


class AttentionPool2d(nn.Module):
    """Kept for state-dict compatibility.  Stage-1 discards its output (model/model_stage1.py:59), so the
    MI355X path never executes it; calling it raises."""

    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim=None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = Linear(embed_dim, embed_dim)
        self.q_proj = Linear(embed_dim, embed_dim)
        self.v_proj = Linear(embed_dim, embed_dim)
        self.c_proj = Linear(embed_dim, output_dim or embed_dim)
        self.num_heads, self.embed_dim, self.spacial_dim = num_heads, embed_dim, spacial_dim
        for p in self.parameters():
            p._tris_no_grad_path = True  # never receives a gradient in Stage-1

    def forward(self, x):
        raise NotImplementedError("AttentionPool2d output is unused by TRIS Stage-1 and is not part of the MI355X path")


class ModifiedResNet(nn.Module):
    def __init__(self, layers, output_dim, heads, input_resolution=224, width=64):
        super().__init__()
        self.output_dim, self.input_resolution = output_dim, input_resolution
        self.conv1 = Conv2d(3, width // 2, 3, stride=2)
        self.bn1 = BatchNorm2d(width // 2)
        self.conv2 = Conv2d(width // 2, width // 2, 3)
        self.bn2 = BatchNorm2d(width // 2)
        self.conv3 = Conv2d(width // 2, width, 3)
        self.bn3 = BatchNorm2d(width)
        self.avgpool = AvgPool2d(2)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)
        self.attnpool = AttentionPool2d(input_resolution // 32, width * 32, heads, output_dim)

    def _make_layer(self, planes, blocks, stride=1):
        layers = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self._inplanes, planes))
        return nn.Sequential(*layers)

    def forward_cl(self, x, hooks=None, taps=True):
        """x [B,3,H,W] (NCHW, as the reference's callers pass it) -> (c1,c2,c3,c4) channels-last [B,h,w,C].
        taps=False: the caller uses c4 only, and feeds it to ONE 1x1 convolution (TRIS.encode_visual): c1..c3 then have no
        consumer but the next stage, which lets the last block of every stage link its bn3 backward like the others.
        hooks: optional {"stem" | "layer1" | "layer2" | "layer3" | "layer4": callable} run right after that stage has been ISSUED -- TRIS uses
        it to issue the text encoder (side stream) in the middle of the trunk, see model_stage1.TRIS.forward."""
        tr = self.training
        if tr:
            ops.h2_auto_step()     # (operand planes: a trunk forward outside any step bracket starts an amax pool of its own)
        x = ops.nchw_to_nhwc(x.float())
        pl = tr and ops.planes_on()    # (h2 with operand planes: see Bottleneck.forward; conv1 reads the image, its gradient stays fp32)
        x = self.conv1(x, stats=tr)                        # (conv1 has Cin=3: not eligible, separate statistics pass)
        x = self.bn1(x, relu=True, lazy=tr and not pl and ops.conv3x3_bnin_ok(x.shape, self.conv2.cout), bwd_link=True,
                     planes=pl)   # folded into conv2 where possible
        x = self.conv2(ops.cut(x), stats=tr)    # (cuts: the stem is the END of backward -- a segmented capture releases each of its
        x = self.bn2(x, relu=True, lazy=tr and not pl and ops.conv3x3_bnin_ok(x.shape, self.conv3.cout), bwd_link=True,
                     planes=pl, dx_planes=pl)   # ... into conv3
        x = ops.cut(x)                          # weight gradients behind its own convolution, not behind the whole stem)
        x = self.bn3(self.conv3(x, stats=tr), relu=True, pool=True, planes=pl, dx_planes=pl)   # bn3 + ReLU + AvgPool2d(2) as one op
        x = ops.cut(x)                                     # (segment boundary of a segmented capture; otherwise x itself)
        if hooks and "stem" in hooks:
            hooks["stem"]()
        outs = []
        red = getattr(self, "grad_reducer", None)  # data-parallel: overlap the gradient all-reduce with backward
        if red is not None:
            x = red.boundary(x, "layer1")          # backward passing this point => every layer1 gradient is written
        for name, layer in (("layer2", self.layer1), ("layer3", self.layer2), ("layer4", self.layer3),
                            ("heads", self.layer4)):
            for k, blk in enumerate(layer):
                x = ops.cut(blk(x, link_out=(not taps) or k + 1 < len(layer)))
            if red is not None:
                x = red.boundary(x, name)          # the boundary AFTER a stage releases the segment of the NEXT one
            outs.append(x)
            stage = {"layer2": "layer1", "layer3": "layer2", "layer4": "layer3", "heads": "layer4"}.get(name)   # the stage just issued
            if hooks and stage in hooks:
                hooks[stage]()
        return tuple(outs)

    def forward(self, x):
        # reference returns NCHW tensors; hand out NCHW-shaped views of the channels-last buffers (no copy)
        outs = [ops.unplanes(o).permute(0, 3, 1, 2) for o in self.forward_cl(x)]   # (h2 operand planes: callers get fp32 values)
        outs.append(None)  # [x_global, x_local] of attnpool: discarded by every Stage-1 caller
        return tuple(outs)


class LayerNorm(nn.Module):
    def __init__(self, w, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(w))
        self.bias = nn.Parameter(torch.zeros(w))

    def forward(self, x, grad_box=None):
        return ops.layer_norm(x, self.weight, self.bias, self.eps, grad_box)


class QuickGELU(nn.Module):
    def forward(self, x):
        return ops.quick_gelu(x)


class MultiheadAttention(nn.Module):
    """Parameter layout of nn.MultiheadAttention (packed in_proj, out_proj); self-attention only."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)
        self.causal = attn_mask is not None

    def forward(self, x):  # x [N, L, W] batch-first (the reference runs sequence-first; same math)
        # x feeds a LayerNorm and, through the residual add fused into the closing Linear, the block output.  The Linear's
        # backward runs first (the LayerNorm's depends on it), leaves its residual gradient in a GradBox, and the
        # LayerNorm backward kernel adds it while writing dX: no separate accumulation pass, no copy.
        use_box = torch.is_grad_enabled() and x.requires_grad and cfg.grad_box
        b1 = ops.GradBox() if use_box else None
        b2 = ops.GradBox() if use_box else None
        h = self.ln_1(x, b1)
        qkv = ops.linear(h, self.attn.in_proj_weight, self.attn.in_proj_bias)
        a = ops.mha(qkv, self.attn.num_heads, self.causal)
        x = self.attn.out_proj(a, resid=x, grad_box_res=b1)
        h = self.ln_2(x, b2)
        if torch.is_grad_enabled() and cfg.mlp_fuse:
            # c_fc + QuickGELU in one launch (the epilogue also stores the pre-activation); the QuickGELU backward rides c_proj's
            # data-gradient epilogue: f has no other consumer
            f = ops.linear_qgelu(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias)
            return ops.linear(f, self.mlp.c_proj.weight, self.mlp.c_proj.bias, x, 0, grad_box_res=b2, act_link=True)
        if torch.is_grad_enabled():
            return self.mlp.c_proj(self.mlp.gelu(self.mlp.c_fc(h)), resid=x, grad_box_res=b2)
        f = self.mlp.c_fc(h, act=2)  # QuickGELU fused into the GEMM epilogue (inference / frozen aux text)
        return self.mlp.c_proj(f, resid=x, grad_box_res=b2)


    def forward_packed(self, x, plan, N, L):
        """the block on PACKED text rows [N L, W] (no gradient; CLIP.encode_text_hidden): every op but the attention is row-wise and
        runs under the row limit of the plan; the attention walks each sentence's own row range"""
        h = self.ln_1(x)
        qkv = ops.linear(h, self.attn.in_proj_weight, self.attn.in_proj_bias)
        a = ops.mha_packed(qkv, plan, N, L, self.attn.num_heads, self.causal)
        x = self.attn.out_proj(a, resid=x)
        h = self.ln_2(x)
        return self.mlp.c_proj(self.mlp.c_fc(h, act=2), resid=x)

    def forward_token0(self, x):
        """The block as the LAST one of a tower that is read at token 0 only (VisionTransformer.forward_patches: ln_post(x[:, 0]) @ proj,
        reference CLIP/clip/model.py:443-446): keys and values need every token, but everything behind the attention is row-wise, and
        only the class token's row of the block output is ever read -- out_proj, ln_2 and the MLP run on [N, W] instead of [N, L, W].
        Same values for that row as forward(x)[:, 0] (row-wise ops; a product of fewer rows may pick another tile: fp32 round-off).
        Returns [N, W].  (No GradBoxes: the residual gradient here belongs to one row of x, autograd adds it.)"""
        h = self.ln_1(x)
        qkv = ops.linear(h, self.attn.in_proj_weight, self.attn.in_proj_bias)
        a = ops.token0(ops.mha(qkv, self.attn.num_heads, self.causal))
        x0 = self.attn.out_proj(a, resid=ops.token0(x))
        h = self.ln_2(x0)
        if torch.is_grad_enabled() and cfg.mlp_fuse:
            f = ops.linear_qgelu(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias)
            return ops.linear(f, self.mlp.c_proj.weight, self.mlp.c_proj.bias, x0, 0, act_link=True)
        if torch.is_grad_enabled():
            return self.mlp.c_proj(self.mlp.gelu(self.mlp.c_fc(h)), resid=x0)
        return self.mlp.c_proj(self.mlp.c_fc(h, act=2), resid=x0)


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x):
        for blk in self.resblocks:
            x = ops.cut(blk(x))        # (segment boundary of a segmented capture of the trunk; otherwise x itself)
        return x

    def forward_token0(self, x):
        """forward(x)[:, 0] without the rows of the last block's output that nobody reads (ResidualAttentionBlock.forward_token0)"""
        blocks = list(self.resblocks)
        for blk in blocks[:-1]:
            x = ops.cut(blk(x))
        return blocks[-1].forward_token0(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        # patch conv keeps the standard contiguous [W,3,ps,ps] layout: it is used as a [W, 3*ps*ps] GEMM operand
        self.conv1 = nn.Module()
        self.conv1.weight = nn.Parameter(torch.randn(width, 3, patch_size, patch_size) * 0.02)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        for blk in self.transformer.resblocks:
            # the tower's Linear weights also exist as operand planes (ops._pl_weight_ok): its products at a few thousand rows and more
            # run on the LDS-DMA loop with the activation converted once per product (ops._a_planes, cfg.gemm_convert)
            for w in (blk.attn.in_proj_weight, blk.attn.out_proj.weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight):
                w._tris_linear_w = True
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.spacial_dim = 7
        # their gradients come back through torch ops (slice / interpolate / cat in forward_spatial, VitAssembleFn's returned
        # dcls / dpos) and are ACCUMULATED by autograd: FusedAdamW.zero_grad clears them each step
        self.class_embedding._tris_accumulates = True
        self.positional_embedding._tris_accumulates = True

    def forward_patches(self, patches):
        """patches [B, G*G, 3*ps*ps] (rows ordered (c,ky,kx), see ops.fg_patches) -> [B, output_dim]"""
        emb = ops.linear(patches, self.conv1.weight)   # [W, 3, ps, ps] contiguous = the [W, 3*ps*ps] GEMM operand; sunk gradient
        x = ops.vit_assemble(emb, self.class_embedding, self.positional_embedding)
        x = self.ln_pre(x)
        if cfg.vit_token0:
            cls = self.ln_post(self.transformer.forward_token0(x))
        else:
            cls = self.ln_post(ops.token0(self.transformer(x)))
        return ops.matmul(cls, self.proj)

    def forward_spatial(self, x):
        """ViT as a dense trunk (BASELINE config 5; the reference sketches it in comments, CLIP/clip/model.py:427-441):
        x [B,3,H,W] -> (cls [B,width], spa [B, H/ps, W/ps, width] channels-last) taken after the last block (no ln_post).
        The spatial positional embedding is bilinearly interpolated from its training grid to the input's grid."""
        B, C, H, Wd = x.shape
        ps = self.patch_size
        ones = torch.ones(B, 1, H, Wd, device=x.device, dtype=torch.float32)
        emb = ops.linear(ops.fg_patches(ones, x.float(), ps), self.conv1.weight)
        gh, gw = H // ps, Wd // ps
        sd = self.input_resolution // ps
        width = self.positional_embedding.shape[1]
        pos = self.positional_embedding
        if (gh, gw) != (sd, sd):
            grid = pos[1:].reshape(1, sd, sd, width).permute(0, 3, 1, 2)                       # [1,width,sd,sd]
            grid = ops.resize_bilinear(grid, (gh, gw), False)                                  # align_corners=False
            pos = torch.cat([pos[0:1], grid.reshape(width, gh * gw).permute(1, 0)], dim=0)
        x = ops.vit_assemble(emb, self.class_embedding, pos.contiguous())
        x = self.ln_pre(x)
        x = self.transformer(x)
        return x[:, 0, :], x[:, 1:, :].reshape(B, gh, gw, width)

    def forward(self, x):
        """x [B,3,R,R] NCHW.  Patch extraction = fg_patches with a unit cam."""
        B, C, R, _ = x.shape
        ones = torch.ones(B, 1, R, R, device=x.device, dtype=torch.float32)
        return self.forward_patches(ops.fg_patches(ones, x.float(), self.patch_size))


class CLIP(nn.Module):
    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 txt_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        self.context_length = context_length
        if isinstance(vision_layers, (tuple, list)):
            self.visual = ModifiedResNet(layers=vision_layers, output_dim=embed_dim, heads=vision_width * 32 // 64,
                                         input_resolution=image_resolution, width=vision_width)
        else:
            self.visual = VisionTransformer(input_resolution=image_resolution, patch_size=vision_patch_size,
                                            width=vision_width, layers=vision_layers, heads=vision_width // 64,
                                            output_dim=embed_dim)
        self.txt_length = txt_length
        self.transformer = Transformer(width=transformer_width, layers=transformer_layers, heads=transformer_heads,
                                       attn_mask=True)  # causal mask applied inside the attention kernel
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.logit_scale._tris_no_grad_path = True
        self.initialize_parameters()

    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        if isinstance(self.visual, ModifiedResNet):
            std = self.visual.attnpool.c_proj.in_features ** -0.5
            for m in (self.visual.attnpool.q_proj, self.visual.attnpool.k_proj, self.visual.attnpool.v_proj,
                      self.visual.attnpool.c_proj):
                nn.init.normal_(m.weight, std=std)
            for layer in (self.visual.layer1, self.visual.layer2, self.visual.layer3, self.visual.layer4):
                for name, param in layer.named_parameters():
                    if name.endswith("bn3.weight"):
                        nn.init.zeros_(param)
        proj_std = (self.transformer.width ** -0.5) * ((2 * self.transformer.layers) ** -0.5)
        attn_std = self.transformer.width ** -0.5
        fc_std = (2 * self.transformer.width) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        """text int [N, L] -> (x [N, L, W], hidden [N, E])"""
        ids = text.long()
        if ids.shape[1] > self.txt_length:
            raise ValueError(f"token length {ids.shape[1]} exceeds txt_length {self.txt_length}")
        red = getattr(self, "grad_reducer", None)
        x = ops.embed(ids, self.token_embedding.weight, self.positional_embedding, reducer=red)
        if red is not None:
            # data-parallel: once backward passes this point every gradient of the text transformer, ln_final and
            # text_projection is written (they were all created after it); the embedding tables follow in finish()
            x = red.boundary(x, "text")
            from ...parallel import TEXT_BOUNDARIES   # sub-segments of the text encoder (see tris_amd.parallel)
            for i, blk in enumerate(self.transformer.resblocks):
                x = blk(x)
                if i in TEXT_BOUNDARIES:
                    x = red.boundary(x, TEXT_BOUNDARIES[i])
        else:
            x = self.transformer(x)
        x = self.ln_final(x)
        hidden = ops.matmul(ops.eot_gather(ids, x), self.text_projection)
        return x, hidden

    def encode_text_hidden(self, text):
        """encode_text(text)[1] -- what every caller of the Stage-1 path uses.  Without gradients (the frozen aux tower: 4 of 5 sentence
        passes of a training step) the pass runs on PACKED rows (cfg.text_pack): `hidden` is the row at each sentence's EOT token and
        the mask is causal, so the positions behind EOT -- 42 % of a RefCOCOg-shaped batch -- cannot reach it
        (tests/test_oracle_golden.py::test_tokens_behind_eot_cannot_reach_hidden) and are not computed.  All extents are device words
        (ops.text_pack_plan): a captured pass replays for any token ids.  Same values as the padded pass up to the tile choice of the
        products (fewer rows)."""
        ids = text.long()
        if (not cfg.text_pack or torch.is_grad_enabled() or not ops.text_packable(ids.shape[1]) or ids.shape[1] > self.txt_length
                or not ids.is_cuda):
            return self.encode_text(text)[1]
        ids = ids.contiguous()
        N, L = ids.shape
        plan = ops.text_pack_plan(ids)
        x = ops.embed_packed(ids, self.token_embedding.weight, self.positional_embedding, plan)
        with ops.rows_limit(plan):
            for blk in self.transformer.resblocks:
                x = blk.forward_packed(x, plan, N, L)
            x = self.ln_final(x)
        return ops.matmul(ops.eot_gather_packed(x, plan, N), self.text_projection)

    def forward(self, image, text):
        raise NotImplementedError("CLIP.forward (zero-shot logits) is not on the Stage-1 path")


def convert_weights(model):
    """fp16 conversion in the reference (CLIP/clip/model.py:583-604); the MI355X path computes in fp32 -> no-op."""
    return model


def build_model(state_dict, txt_length=40):
    vit = "visual.proj" in state_dict
    if vit:
        vision_width = state_dict["visual.conv1.weight"].shape[0]
        vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
        grid = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        image_resolution = vision_patch_size * grid
    else:
        counts = [len(set(k.split(".")[2] for k in state_dict if k.startswith(f"visual.layer{b}"))) for b in (1, 2, 3, 4)]
        vision_layers = tuple(counts)
        vision_width = state_dict["visual.layer1.0.conv1.weight"].shape[0]
        out_w = round((state_dict["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        vision_patch_size = None
        assert out_w ** 2 + 1 == state_dict["visual.attnpool.positional_embedding"].shape[0]
        image_resolution = out_w * 32
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    width = state_dict["ln_final.weight"].shape[0]
    layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 txt_length, vocab_size, width, width // 64, layers)
    sd = {k: v.float() for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    model.load_state_dict(sd, strict=False)
    return model.eval()


# constructor arguments of the CLIP variants used by Stage-1 (no checkpoint needed to build the architecture)
ARCH = {
    "RN50": dict(embed_dim=1024, image_resolution=224, vision_layers=(3, 4, 6, 3), vision_width=64,
                 vision_patch_size=None, context_length=77, vocab_size=49408, transformer_width=512,
                 transformer_heads=8, transformer_layers=12),
    "RN101": dict(embed_dim=512, image_resolution=224, vision_layers=(3, 4, 23, 3), vision_width=64,
                  vision_patch_size=None, context_length=77, vocab_size=49408, transformer_width=512,
                  transformer_heads=8, transformer_layers=12),
    "ViT-B/32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                     context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                     transformer_layers=12),
    "ViT-B/16": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=16,
                     context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8,
                     transformer_layers=12),
}
ARCH["ViT-B-32"] = ARCH["ViT-B/32"]  # train_stage1.py:167 spells it with a dash
