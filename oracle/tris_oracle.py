"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the TRIS Stage-1 hot path.

A plain-PyTorch, fp32, CPU restatement of the reference algorithm, written functionally over
a flat state dict (name -> tensor) instead of as nn.Modules.  It is the *checker* for the HIP
path (tests/, __graft_entry__.smoke) and the timed `cpu_baseline` leg of bench.py; the product
package `tris_amd` never imports it.

Parity pin: `oracle/gen_golden.py` runs this file side by side with the real reference imported
from /root/reference (oracle/ref_shim.py) on seed-filled weights and writes tests/golden/*.npz;
tests/test_oracle_golden.py re-checks this file against those committed vectors on any machine,
and tests/test_oracle_vs_reference.py re-checks against the live reference when it is present.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# encoders (CLIP/clip/model.py)
# --------------------------------------------------------------------------------------


def _ln(sd, p, x):
    # LayerNorm subclass computing in fp32, eps 1e-5 -- CLIP/clip/model.py:352-358
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mha(sd, p, x, heads, mask):
    """nn.MultiheadAttention self-attention, batch-first restatement.
    CLIP/clip/model.py:369 (module), :380-382 (call with additive attn_mask)."""
    N, L, W = x.shape
    d = W // heads
    qkv = x @ sd[p + ".in_proj_weight"].t() + sd[p + ".in_proj_bias"]
    q, k, v = qkv.split(W, dim=-1)
    q = q.view(N, L, heads, d).transpose(1, 2)
    k = k.view(N, L, heads, d).transpose(1, 2)
    v = v.view(N, L, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    if mask is not None:
        s = s + mask
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(N, L, W)
    return o @ sd[p + ".out_proj.weight"].t() + sd[p + ".out_proj.bias"]


def _resblock(sd, p, x, heads, mask):
    # ResidualAttentionBlock.forward -- CLIP/clip/model.py:384-387; QuickGELU :361-363
    x = x + _mha(sd, p + ".attn", _ln(sd, p + ".ln_1", x), heads, mask)
    h = _ln(sd, p + ".ln_2", x) @ sd[p + ".mlp.c_fc.weight"].t() + sd[p + ".mlp.c_fc.bias"]
    h = h * torch.sigmoid(1.702 * h)
    return x + (h @ sd[p + ".mlp.c_proj.weight"].t() + sd[p + ".mlp.c_proj.bias"])


def _n_layers(sd, p):
    i = 0
    while f"{p}.resblocks.{i}.ln_1.weight" in sd:
        i += 1
    return i


def encode_text(sd, p, ids):
    """CLIP.encode_text -- CLIP/clip/model.py:552-564 (causal mask: :537-543).
    ids int64 [N,L] -> (x [N,L,W], hidden [N,E]).  `p` is the CLIP prefix, e.g. 'backbone.'."""
    N, L = ids.shape
    x = sd[p + "token_embedding.weight"][ids] + sd[p + "positional_embedding"][:L]
    W = x.shape[-1]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    for i in range(_n_layers(sd, p + "transformer")):
        x = _resblock(sd, f"{p}transformer.resblocks.{i}", x, W // 64, mask)
    x = _ln(sd, p + "ln_final", x)
    hidden = x[torch.arange(N), ids.argmax(dim=-1)] @ sd[p + "text_projection"]
    return x, hidden


def _bn(sd, p, x, train):
    # nn.BatchNorm2d, momentum 0.1, eps 1e-5; batch statistics when train (train_stage1.py:288)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], train, 0.1, 1e-5)


def _bottleneck(sd, p, x, stride, train):
    # Bottleneck.forward -- CLIP/clip/model.py:42-55 (stride = avgpool after conv2, :25, :36-40)
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"]), train))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1), train))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]), train)
    if (p + ".downsample.0.weight") in sd:
        idn = F.avg_pool2d(x, stride) if stride > 1 else x
        idn = _bn(sd, p + ".downsample.1", F.conv2d(idn, sd[p + ".downsample.0.weight"]), train)
    else:
        idn = x
    return F.relu(out + idn)


def attnpool(sd, p, x, heads=32, spacial_dim=7):
    """AttentionPool2d.forward -- CLIP/clip/model.py:70-104.  Result is discarded by TRIS
    (model/model_stage1.py:59) and only executed in `faithful` timing mode."""
    B, C, H, W = x.shape
    t = x.reshape(B, C, H * W).permute(0, 2, 1)
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1)  # [B, HW+1, C]
    pe = sd[p + ".positional_embedding"]
    sp = F.interpolate(pe[1:].reshape(1, spacial_dim, spacial_dim, C).permute(0, 3, 1, 2),
                       size=(H, W), mode="bilinear")
    pos = torch.cat([pe[0:1], sp.reshape(C, H * W).permute(1, 0)], dim=0)
    t = t + pos[None]
    d = C // heads
    q = (t @ sd[p + ".q_proj.weight"].t() + sd[p + ".q_proj.bias"]).view(B, -1, heads, d).transpose(1, 2)
    k = (t @ sd[p + ".k_proj.weight"].t() + sd[p + ".k_proj.bias"]).view(B, -1, heads, d).transpose(1, 2)
    v = (t @ sd[p + ".v_proj.weight"].t() + sd[p + ".v_proj.bias"]).view(B, -1, heads, d).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, H * W + 1, C)
    o = o @ sd[p + ".c_proj.weight"].t() + sd[p + ".c_proj.bias"]
    return o[:, 0], o[:, 1:].permute(0, 2, 1).reshape(B, -1, H, W)


def encode_image_rn(sd, p, x, train, with_attnpool=False, layers=(3, 4, 6, 3)):
    """ModifiedResNet.forward -- CLIP/clip/model.py:254-279.  Returns (c1,c2,c3,c4[,pool])."""
    v = p + "visual"
    x = F.relu(_bn(sd, v + ".bn1", F.conv2d(x, sd[v + ".conv1.weight"], stride=2, padding=1), train))
    x = F.relu(_bn(sd, v + ".bn2", F.conv2d(x, sd[v + ".conv2.weight"], padding=1), train))
    x = F.relu(_bn(sd, v + ".bn3", F.conv2d(x, sd[v + ".conv3.weight"], padding=1), train))
    x = F.avg_pool2d(x, 2)
    outs = []
    for li, n in enumerate(layers):
        for b in range(n):
            stride = 2 if (li > 0 and b == 0) else 1
            x = _bottleneck(sd, f"{v}.layer{li + 1}.{b}", x, stride, train)
        outs.append(x)
    if with_attnpool:
        outs.append(attnpool(sd, v + ".attnpool", x))
    return tuple(outs)


def encode_image_vit(sd, p, x):
    """VisionTransformer.forward -- CLIP/clip/model.py:419-448.  x [N,3,224,224] -> [N,512]."""
    v = p + "visual"
    w = sd[v + ".conv1.weight"]
    x = F.conv2d(x, w, stride=w.shape[-1])
    N, Wd = x.shape[:2]
    x = x.reshape(N, Wd, -1).permute(0, 2, 1)
    cls = sd[v + ".class_embedding"].expand(N, 1, Wd)
    x = torch.cat([cls, x], dim=1) + sd[v + ".positional_embedding"]
    x = _ln(sd, v + ".ln_pre", x)
    for i in range(_n_layers(sd, v + ".transformer")):
        x = _resblock(sd, f"{v}.transformer.resblocks.{i}", x, Wd // 64, None)
    x = _ln(sd, v + ".ln_post", x[:, 0])
    return x @ sd[v + ".proj"]


def encode_image_vit_spatial(sd, p, x):
    """The dense-trunk variant the reference sketches in comments (CLIP/clip/model.py:427-441): interpolated spatial
    positional embedding, spatial tokens after the last block reshaped to [N, width, H/ps, W/ps] (no ln_post).
    PARITY UNPINNED as a whole (the reference never runs it); built only from pieces that are pinned (G1/G2)."""
    v = p + "visual"
    w = sd[v + ".conv1.weight"]
    x = F.conv2d(x, w, stride=w.shape[-1])
    N, Wd, H, W = x.shape
    x = x.reshape(N, Wd, -1).permute(0, 2, 1)
    cls = sd[v + ".class_embedding"].expand(N, 1, Wd)
    x = torch.cat([cls, x], dim=1)
    pos = sd[v + ".positional_embedding"]
    sdim = int(round(math.sqrt(pos.shape[0] - 1)))
    spatial = F.interpolate(pos[1:].reshape(1, sdim, sdim, Wd).permute(0, 3, 1, 2), size=(H, W), mode="bilinear")
    pos = torch.cat([pos[0:1], spatial.reshape(Wd, H * W).permute(1, 0)], dim=0)
    x = _ln(sd, v + ".ln_pre", x + pos)
    for i in range(_n_layers(sd, v + ".transformer")):
        x = _resblock(sd, f"{v}.transformer.resblocks.{i}", x, Wd // 64, None)
    return x[:, 0, :], x[:, 1:, :].permute(0, 2, 1).reshape(N, Wd, H, W)


# --------------------------------------------------------------------------------------
# Stage-1 model (model/model_stage1.py, model/attn.py)
# --------------------------------------------------------------------------------------


def _conv_in_relu(sd, p, x, relu=True):
    # nn.Sequential(Conv2d 1x1, InstanceNorm2d(affine), [ReLU]) -- model/attn.py:73-87, 104-107
    y = F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"])
    y = F.instance_norm(y, None, None, sd[p + ".1.weight"], sd[p + ".1.bias"], True, 0.1, 1e-5)
    return F.relu(y) if relu else y


def bilateral_prompt(sd, p, vis, lan):
    """bilateral_prompt.forward -- model/attn.py:111-136.
    vis [B,C,h,w], lan [B,C,N] -> (new_vis [B,C,h,w], new_lan [B,N,C])."""
    B, C, H, W = vis.shape
    lan = lan.transpose(1, 2)
    Ci = lan.shape[-1]
    Qv, Kv, Vv = (_conv_in_relu(sd, p + f".v_proj{i}", vis) for i in (1, 2, 3))
    Qt, Kt, Vt = (F.relu(lan @ sd[p + f".t_proj{i}.0.weight"].t() + sd[p + f".t_proj{i}.0.bias"])
                  for i in (1, 2, 3))
    Qv = Qv.reshape(B, C, -1).transpose(1, 2)
    Av = torch.softmax(Qv @ Kt.transpose(1, 2) / math.sqrt(Ci), dim=2)
    At = torch.softmax(Qt @ Kv.reshape(B, C, -1) / math.sqrt(Ci), dim=2)
    new_vis = (Av @ Vt).permute(0, 2, 1).reshape(B, C, H, W)
    new_lan = At @ Vv.reshape(B, C, -1).transpose(1, 2)
    new_vis = _conv_in_relu(sd, p + ".v_output", new_vis, relu=False)
    new_lan = new_lan @ sd[p + ".t_output.0.weight"].t() + sd[p + ".t_output.0.bias"]
    return new_vis, new_lan


def pixel_attention(sd, p, vis, lan):
    """PixelAttention.forward (Stage-2 pixel x token cross attention) -- model/attn.py:37-65.
    vis [N,Ci,H,W], lan [N,Ct,T] -> [N,Ci,H,W]."""
    N, Ci, H, W = vis.shape
    Lk = F.conv1d(lan, sd[p + ".Wk.weight"], sd[p + ".Wk.bias"])
    Lv = F.conv1d(lan, sd[p + ".Wv.weight"], sd[p + ".Wv.bias"])
    Vq = F.instance_norm(F.conv2d(vis, sd[p + ".Wq.weight"], sd[p + ".Wq.bias"]), weight=sd[p + ".ins_q.weight"],
                         bias=sd[p + ".ins_q.bias"], eps=1e-5)
    Vq = Vq.view(N, Ci, H * W).permute(0, 2, 1)
    attn = torch.softmax(Vq.matmul(Lk) / math.sqrt(Ci), dim=2)
    G = attn.matmul(Lv.permute(0, 2, 1)).permute(0, 2, 1).reshape(N, Ci, H, W)
    Gi = F.instance_norm(F.conv2d(G, sd[p + ".Ww.weight"], sd[p + ".Ww.bias"]), weight=sd[p + ".ins_w.weight"],
                         bias=sd[p + ".ins_w.bias"], eps=1e-5)
    Vo = F.relu(F.conv2d(vis, sd[p + ".Wm.weight"], sd[p + ".Wm.bias"]))
    return F.relu(F.conv2d(Vo * Gi, sd[p + ".Wo.weight"], sd[p + ".Wo.bias"]))


def tris_heads(sd, c4, hidden, size, train, focal_p=3.0, focal_c=0.01, attn_multi=0.1, return_score=False):
    """TRIS.forward behind the two encoders -- model/model_stage1.py:61-119: projections, L2 norms, bilateral_prompt,
    score, cls head, response maps.  c4 [B,Cv,h,w], hidden [N,E] with N = B sentences (one per image)."""
    B = c4.shape[0]
    lan = hidden @ sd["lan_project.weight"].t() + sd["lan_project.bias"]
    vis = F.conv2d(c4, sd["vis_project.weight"], sd["vis_project.bias"])
    h_, w_ = vis.shape[2:]
    vis_t = vis.flatten(2).transpose(1, 2)
    lan = lan.unsqueeze(0).repeat(B, 1, 1)
    nv = vis_t / vis_t.norm(dim=-1, keepdim=True)
    nl = lan / lan.norm(dim=-1, keepdim=True)
    if attn_multi > 0:
        new_vis, new_lan = bilateral_prompt(sd, "attn_fusion", nv.permute(0, 2, 1).reshape(B, -1, h_, w_),
                                            nl.transpose(1, 2))
        nv = new_vis.flatten(2).transpose(1, 2) * 0.1 + nv
        nl = new_lan * 0.1 + nl
    ls = sd["logit_scale"].exp()
    score = ls * torch.bmm(nv, nl.transpose(1, 2))  # [B(img), P, B(sent)]
    cls_out = cls_fg = None
    if train:
        st = score.transpose(1, 2).reshape(B, -1, h_, w_)
        st = torch.cat([torch.ones_like(st[:, :1]), st], 1)
        masks = torch.softmax(st, dim=1).view(B, B + 1, -1)
        feats = st.view(B, B + 1, -1)
        cls_1 = feats.mean(-1) + feats.max(dim=-1).values
        mm = masks.mean(-1)
        cls_2 = torch.pow(1 - mm, focal_p) * torch.log(focal_c + mm)
        cls_out = cls_1[:, 1:] + cls_2[:, 1:]
        cls_fg = torch.diagonal(mm[:, 1:])
    diag = torch.stack([score[i, :, i].view(1, h_, w_) for i in range(B)], 0)
    seg = F.interpolate(diag, size=size, mode="bilinear", align_corners=False)
    if return_score:
        return diag, seg
    if train:
        return cls_out, cls_fg, F.relu(seg), torch.sigmoid(seg), ls
    return F.relu(seg)


def tris_forward(sd, img, word_id, train, focal_p=3.0, focal_c=0.01, attn_multi=0.1,
                 with_attnpool=False, return_score=False, vit_trunk=False):
    """TRIS.forward -- model/model_stage1.py:54-119 (focal_loss :122-123, Upsample model/utils.py:5-10)."""
    _, hidden = encode_text(sd, "backbone.", word_id)
    if vit_trunk:
        c4 = encode_image_vit_spatial(sd, "backbone.", img)[1]
    else:
        c4 = encode_image_rn(sd, "backbone.", img, train, with_attnpool)[3]
    return tris_heads(sd, c4, hidden, img.shape[2:], train, focal_p, focal_c, attn_multi, return_score)


# --------------------------------------------------------------------------------------
# loss block + step (train_stage1.py)
# --------------------------------------------------------------------------------------


def clip_forward(aux, images, tokens):
    """clip_forward -- train_stage1.py:263-278 (= loss/clip_loss.py:5-20).  aux = ViT-B/32 state dict."""
    f_i = encode_image_vit(aux, "", images)
    f_t = encode_text(aux, "", tokens)[1]
    f_i = f_i / f_i.norm(dim=-1, keepdim=True)
    f_t = f_t / f_t.norm(dim=-1, keepdim=True)
    return (f_i * f_t).sum(-1).view(-1, 1, 1)


def max_loss(x):
    # MaxLoss -- train_stage1.py:280-284
    return -(torch.log(x.clamp(0.0001, 0.9999))).mean()


def stage1_loss_block(aux, img, ids, neg, cls, sig, w=(1.0, 5.0, 2.0), faithful=False):
    """The loss block behind TRIS.forward -- train_stage1.py:327-364: fg construction, CLIP-guided fg loss (l1),
    negative-sample loss (l5), cls loss (l4), weighted total."""
    B = img.shape[0]
    cam = F.interpolate(sig, (224, 224), mode="bilinear", align_corners=True)
    im224 = F.interpolate(img, (224, 224), mode="bilinear", align_corners=True)
    fg = cam * im224
    if faithful:
        l1 = max_loss(clip_forward(aux, fg, ids))
        f_i = encode_image_vit(aux, "", fg)
    else:
        f_i = encode_image_vit(aux, "", fg)
        f_t = encode_text(aux, "", ids)[1]
        l1 = max_loss(((f_i / f_i.norm(dim=-1, keepdim=True)) *
                       (f_t / f_t.norm(dim=-1, keepdim=True))).sum(-1))
    l5 = torch.zeros(())
    if neg is not None:
        fn = f_i / f_i.norm(dim=-1, keepdim=True)
        if faithful:  # train_stage1.py:346-353, one tiny text forward per image
            for i in range(B):
                t = encode_text(aux, "", neg[i])[1]
                t = t / t.norm(dim=-1, keepdim=True)
                l5 = l5 + (-(torch.log(1 - fn[i:i + 1] @ t.t())).mean())
            l5 = l5 / B
        else:
            K = neg.shape[1]
            t = encode_text(aux, "", neg.reshape(B * K, -1))[1]
            t = (t / t.norm(dim=-1, keepdim=True)).view(B, K, -1)
            l5 = (-(torch.log(1 - (fn[:, None] * t).sum(-1)))).mean(1).mean()
    l4 = F.multilabel_soft_margin_loss(cls, torch.eye(B))
    loss = w[0] * l1 + w[1] * l4 + w[2] * l5
    return loss, l1, l4, l5


def stage1_losses(sd, aux, batch, w=(1.0, 5.0, 2.0), faithful=False):
    """One Stage-1 forward + loss block -- train_stage1.py:317-364.

    faithful=True mirrors the reference's redundant work (attnpool, second aux image forward,
    per-image negative-text loop); faithful=False computes the same numbers once ("lean",
    SURVEY.md §6: identical loss).  Returns dict(loss, l1, l4, l5, cls, sig)."""
    img, ids = batch["img"], batch["word_ids"]
    cls, cls_fg, relu_map, sig, _ = tris_forward(sd, img, ids, True, with_attnpool=faithful)
    loss, l1, l4, l5 = stage1_loss_block(aux, img, ids, batch.get("neg_word_ids"), cls, sig, w, faithful)
    return {"loss": loss, "l1": l1, "l4": l4, "l5": l5, "cls": cls, "cls_fg": cls_fg,
            "sig": sig, "relu": relu_map}


def stage1_losses_ddp(sd, aux, batch, world, w=(1.0, 5.0, 2.0)):
    """N-rank correctness oracle of the data-parallel step (reference: DistributedDataParallel +
    SyncBatchNorm, train_stage1.py:69-70): ONE process on the concatenated shards.  The only cross-rank couplings of the
    reference's step are (1) SyncBatchNorm -- batch statistics over all world*B images, which is what train-mode BN over
    the concatenated batch computes (running stats included) -- and (2) the gradient MEAN over ranks, which is the
    gradient of the mean of the per-rank losses.  Everything behind the two encoders is rank-local: each rank contrasts
    its B images with its own B sentences only (block-diagonal cls labels), model/model_stage1.py:66,107.
    batch: the shards concatenated in rank order.  Returns dict(loss = mean over ranks, per_rank = [world][4] tensors,
    cls / sig = per-rank lists)."""
    img, ids, neg = batch["img"], batch["word_ids"], batch.get("neg_word_ids")
    n = img.shape[0] // world
    _, hidden = encode_text(sd, "backbone.", ids)
    c4 = encode_image_rn(sd, "backbone.", img, True)[3]          # BN over world*B images = SyncBatchNorm
    per, cls_l, sig_l = [], [], []
    for r in range(world):
        sl = slice(r * n, (r + 1) * n)
        cls, _, _, sig, _ = tris_heads(sd, c4[sl], hidden[sl], img.shape[2:], True)
        per.append(stage1_loss_block(aux, img[sl], ids[sl], None if neg is None else neg[sl], cls, sig, w))
        cls_l.append(cls)
        sig_l.append(sig)
    loss = sum(p[0] for p in per) / world
    return {"loss": loss, "per_rank": per, "cls": cls_l, "sig": sig_l}


def backbone_keys(sd):
    return [k for k in sd if k.startswith("backbone.")]


def trainable_split(sd):
    """TRIS.trainable_parameters -- model/model_stage1.py:44-52: (backbone params, new-head params);
    the top-level `logit_scale` is in neither group.  Buffers excluded."""
    def is_buf(k):
        return k.endswith(("running_mean", "running_var", "num_batches_tracked"))
    bb = [k for k in sd if k.startswith("backbone.") and not is_buf(k)]
    new = [k for k in sd if k.startswith(("vis_project.", "lan_project.", "attn_fusion."))]
    return bb, new


def adamw_step(params, grads, state, lr, wd=0.01, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW single step (train_stage1.py:135-139, 370), written out."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    for k in params:
        g = grads.get(k)
        if g is None:
            continue
        p = params[k]
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        p.mul_(1 - lr * wd)
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = (v.sqrt() / math.sqrt(1 - betas[1] ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - betas[0] ** t))


def train_step(sd, aux, batch, lr=5e-5, lr_multi=0.1, wd=0.01, state=None, faithful=False,
               aux_weight_grads=None):
    """zero_grad / backward / AdamW step -- train_stage1.py:364-372.  Mutates `sd` in place.
    Returns (losses dict of floats, grads dict)."""
    state = {} if state is None else state
    bb, new = trainable_split(sd)
    # attnpool + backbone.logit_scale never receive a gradient (unused outputs)
    leaves = [k for k in bb + new + ["logit_scale"]]
    for k in leaves:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    want_aux = faithful if aux_weight_grads is None else aux_weight_grads
    if want_aux:  # the reference leaves requires_grad=True on the frozen aux CLIP (train_stage1.py:168)
        for k, t in aux.items():
            if t.is_floating_point():
                t.requires_grad_(True)
                t.grad = None
    out = stage1_losses(sd, aux, batch, faithful=faithful)
    out["loss"].backward()
    grads = {k: sd[k].grad for k in leaves if sd[k].grad is not None}
    with torch.no_grad():
        for k in leaves:
            sd[k].requires_grad_(False)
        adamw_step({k: sd[k] for k in bb}, grads, state.setdefault("bb", {}), lr * lr_multi, wd)
        adamw_step({k: sd[k] for k in new}, grads, state.setdefault("new", {}), lr, wd)
    if want_aux:
        for t in aux.values():
            t.requires_grad_(False)
            t.grad = None
    return {k: float(out[k].detach()) for k in ("loss", "l1", "l4", "l5")}, grads


# --------------------------------------------------------------------------------------
# evaluation post-processing (validate.py)
# --------------------------------------------------------------------------------------


def eval_postprocess(relu_map, target):
    """validate.py:180-190 + utils/util.py:9-15 for ONE (image, sentence).
    relu_map [1,1,S,S], target bool/int [oH,oW] -> (I, U, pred mask, normalised cam)."""
    oH, oW = target.shape
    pred = F.interpolate(relu_map, (oH, oW), mode="bilinear", align_corners=True)[0]
    pred = pred / (pred.amax(dim=(1, 2), keepdim=True) + 1e-5)
    cam = pred[0]
    mask = cam > 1e-9
    t = target.bool()
    I = int((mask & t).sum())
    U = int((mask | t).sum())
    return I, U, mask, cam


def hit_test(cam, boxes, gt_mask):
    """isCorrectHit -- validate.py:106-117: arg-max point inside any GT box / on the GT mask."""
    idx = int(torch.argmax(cam))
    y, x = divmod(idx, cam.shape[1])
    hitm = 1 if bool(gt_mask[y, x]) else 0
    for b in boxes:
        if b[0] <= x <= b[2] and b[1] <= y <= b[3]:
            return 1, (y, x), hitm
    return 0, (y, x), hitm


def prms_scores(sd, aux, img, ids_all):
    """validate_same_sentence scoring -- validate.py:299-333 for ONE ref.
    img [1,3,S,S], ids_all [S_ref, L] -> (maps [S_ref,1,H,W], score[S_ref]); the reference keeps the first arg-max."""
    maps = torch.cat([tris_forward(sd, img, ids_all[j:j + 1], False) for j in range(ids_all.shape[0])], 0)
    im = F.interpolate(img, (224, 224), mode="bilinear", align_corners=True)
    cam = F.interpolate(maps, (224, 224), mode="bilinear", align_corners=True)
    f_i = encode_image_vit(aux, "", cam * im)
    f_t = encode_text(aux, "", ids_all)[1]
    f_i = f_i / f_i.norm(dim=-1, keepdim=True)
    f_t = f_t / f_t.norm(dim=-1, keepdim=True)
    return maps, (f_i @ f_t.t()).sum(1)
