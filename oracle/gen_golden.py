"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz FROM THE REAL REFERENCE.

Run in the build container (where /root/reference exists):

    python oracle/gen_golden.py

It imports fawnliu/TRIS through oracle/ref_shim.py, seed-fills its weights
(tris_amd.utils.synth.seed_fill, seed 1234 / 4321), feeds the synthetic batch
(tris_amd.utils.synth.synthetic_batch) and records small outputs.  The vectors are data
(inputs are regenerated from seeds; expected outputs are stored); no reference source travels.
It also prints the max deviation of oracle/tris_oracle.py from the reference on each vector.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from oracle import tris_oracle as O  # noqa: E402
from tris_amd.utils.synth import seed_fill, synthetic_batch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PROBES = [
    "vis_project.weight", "vis_project.bias", "lan_project.weight",
    "attn_fusion.v_proj1.0.weight", "attn_fusion.v_proj2.1.weight", "attn_fusion.t_proj3.0.weight",
    "attn_fusion.v_output.1.bias", "attn_fusion.t_output.0.bias",
    "backbone.visual.conv1.weight", "backbone.visual.bn1.weight", "backbone.visual.layer1.0.conv2.weight",
    "backbone.visual.layer2.0.downsample.0.weight", "backbone.visual.layer4.2.conv3.weight",
    "backbone.visual.layer4.2.bn3.bias",
    "backbone.transformer.resblocks.0.attn.in_proj_weight", "backbone.transformer.resblocks.11.mlp.c_fc.bias",
    "backbone.transformer.resblocks.5.ln_1.weight", "backbone.positional_embedding",
    "backbone.text_projection", "backbone.ln_final.weight", "logit_scale",
]


def dev(a, b):
    return float((a.detach() - b.detach()).abs().max())


def crop(t, n=16):
    return t[..., :n, :n].detach().numpy().copy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = ref_shim.make_tris()
    seed_fill(ref.state_dict(), 1234)  # state_dict tensors alias the parameters
    aux = ref_shim.make_aux_clip(20)
    seed_fill(aux.state_dict(), 4321)
    keys = list(ref.state_dict().keys())
    with open(os.path.join(OUT, "state_dict_keys.txt"), "w") as f:
        for k, v in ref.state_dict().items():
            f.write(f"{k} {list(v.shape)}\n")
    with open(os.path.join(OUT, "aux_state_dict_keys.txt"), "w") as f:
        for k, v in aux.state_dict().items():
            f.write(f"{k} {list(v.shape)}\n")
    print("TRIS keys", len(keys), "aux keys", len(aux.state_dict()))

    def osd(m):
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    # ---- G1 text encoder, G2 image encoder -------------------------------------------------
    batch = synthetic_batch(2, 320, 20, 3, seed=7)
    img, ids, neg = batch["img"], batch["word_ids"], batch["neg_word_ids"]
    g = {}
    with torch.no_grad():
        ref.eval()
        x, hidden = ref.backbone.encode_text(ids)
        g["text_hidden"] = hidden.numpy()
        g["text_x_sum"] = np.array([float(x.sum()), float(x.abs().sum())])
        sd = osd(ref)
        ox, oh = O.encode_text(sd, "backbone.", ids)
        print("G1 text   oracle-vs-ref", dev(oh, hidden), dev(ox, x))
        for mode in ("eval", "train"):
            ref.train(mode == "train")
            sd = osd(ref)  # fresh running stats copy
            c = ref.backbone.encode_image(img)
            oc = O.encode_image_rn(sd, "backbone.", img, mode == "train")
            for i in range(4):
                g[f"c{i + 1}_{mode}_crop"] = c[i][:, :8, :8, :8].numpy().copy()
                g[f"c{i + 1}_{mode}_stat"] = np.array([float(c[i].mean()), float(c[i].std()), float(c[i].abs().max())])
                print(f"G2 c{i + 1} {mode} oracle-vs-ref", dev(oc[i], c[i]))
            if mode == "train":
                g["bn1_running_mean_after"] = ref.backbone.visual.bn1.running_mean.numpy().copy()
                g["l4_bn3_running_var_after"] = ref.backbone.visual.layer4[2].bn3.running_var.numpy().copy()
                print("G2 running stats", dev(sd["backbone.visual.bn1.running_mean"], ref.backbone.visual.bn1.running_mean),
                      dev(sd["backbone.visual.layer4.2.bn3.running_var"], ref.backbone.visual.layer4[2].bn3.running_var))
        # restore running stats by re-filling (train pass above updated them)
        seed_fill(ref.state_dict(), 1234)
    np.savez_compressed(os.path.join(OUT, "g1_g2_encoders.npz"), **g)

    # ---- G3 bilateral_prompt -------------------------------------------------------------
    g = {}
    gen = torch.Generator().manual_seed(11)
    for B in (1, 2, 4):
        vis = torch.randn(B, 1024, 10, 10, generator=gen)
        vis = vis / vis.norm(dim=1, keepdim=True)
        lan = torch.randn(B, 1024, B, generator=gen)
        lan = lan / lan.norm(dim=1, keepdim=True)
        with torch.no_grad():
            nv, nl = ref.attn_fusion(vis, lan)
            onv, onl = O.bilateral_prompt(osd(ref), "attn_fusion", vis, lan)
        print(f"G3 B={B} oracle-vs-ref", dev(onv, nv), dev(onl, nl))
        g[f"B{B}_new_lan"] = nl.numpy()
        g[f"B{B}_new_vis_crop"] = nv[:, :64].numpy().copy()
        g[f"B{B}_new_vis_sum"] = np.array([float(nv.sum()), float(nv.abs().sum())])
    np.savez_compressed(os.path.join(OUT, "g3_bilateral_prompt.npz"), **g)

    # ---- G4 TRIS.forward (eval B=1,2 ; train B=2) --------------------------------------
    g = {}
    with torch.no_grad():
        ref.eval()
        for B in (1, 2):
            o = ref(img[:B], ids[:B])
            oo = O.tris_forward(osd(ref), img[:B], ids[:B], False)
            print(f"G4 eval B={B} oracle-vs-ref", dev(oo, o))
            g[f"eval_B{B}_crop"] = crop(o)
            g[f"eval_B{B}_stat"] = np.array([float(o.mean()), float(o.max()), float((o > 0).float().mean())])
            # the pre-upsample 10x10 diagonal maps: average-pool of nothing -- recover via oracle layout
            d, _ = O.tris_forward(osd(ref), img[:B], ids[:B], False, return_score=True)
            g[f"eval_B{B}_diag10"] = d.numpy()
            g[f"eval_B{B}_full_ds4"] = o[:, :, ::4, ::4].numpy().copy()
    ref.train()
    sd = osd(ref)
    with torch.no_grad():
        cls, fgc, r, s, ls = ref(img, ids)
        ocls, ofg, orr, os_, ols = O.tris_forward(sd, img, ids, True)
        print("G4 train oracle-vs-ref", dev(ocls, cls), dev(ofg, fgc), dev(orr, r), dev(os_, s), dev(ols, ls))
        g.update(train_cls_out=cls.numpy(), train_cls_fg=fgc.numpy(), train_relu_crop=crop(r), train_sig_crop=crop(s),
                 train_relu_ds4=r[:, :, ::4, ::4].numpy().copy(), train_sig_ds4=s[:, :, ::4, ::4].numpy().copy(),
                 train_logit_scale=np.array(float(ls)))
    seed_fill(ref.state_dict(), 1234)
    np.savez_compressed(os.path.join(OUT, "g4_tris_forward.npz"), **g)

    # ---- G5 losses, G6 gradient probes + params after one AdamW step -----------------------
    g = {}
    ref.train()
    aux.eval()
    sd = osd(ref)
    auxsd = {k: v.detach().clone() for k, v in aux.state_dict().items()}
    bb, new = ref.trainable_parameters()
    opt = torch.optim.AdamW([{"params": bb, "lr": 5e-5 * 0.1}, {"params": new, "lr": 5e-5}], lr=5e-5, weight_decay=0.01)
    cls, _, _, sig, _ = ref(img, ids)
    cam = F.interpolate(sig, (224, 224), mode="bilinear", align_corners=True)
    im = F.interpolate(img, (224, 224), mode="bilinear", align_corners=True)
    fg = torch.stack([cam[i] * im[i] for i in range(2)], 0)
    from loss.clip_loss import clip_forward  # reference loss/clip_loss.py:5
    x = clip_forward(aux, fg, ids)
    l1 = -(torch.log(x.clamp(0.0001, 0.9999))).mean()
    f_i = aux.encode_image(fg)
    l5 = torch.zeros(())
    for i in range(2):
        _, t = aux.encode_text(neg[i])
        fi = f_i[i].reshape(1, -1)
        fi = fi / fi.norm(dim=-1, keepdim=True)
        t = t / t.norm(dim=-1, keepdim=True)
        l5 = l5 + (-(torch.log(1 - fi @ t.t())).mean())
    l5 = l5 / 2
    l4 = F.multilabel_soft_margin_loss(cls, torch.eye(2))
    loss = l1 * 1 + l4 * 5 + l5 * 2
    opt.zero_grad()
    loss.backward()
    named = dict(ref.named_parameters())
    g["losses"] = np.array([float(loss), float(l1), float(l4), float(l5)])
    g["fg_cos"] = x.detach().numpy().reshape(-1)
    for k in PROBES:
        gr = named[k].grad
        g["grad_norm." + k] = np.array(float(gr.norm()))
        g["grad_head." + k] = gr.reshape(-1)[:16].detach().numpy().copy()
    used = torch.unique(ids)
    g["grad_tok_rows"] = named["backbone.token_embedding.weight"].grad[used][:, :8].numpy().copy()
    g["grad_tok_ids"] = used.numpy()
    nograd = [k for k, p in named.items() if p.grad is None]
    g["nograd_keys"] = np.array(nograd)
    opt.step()
    for k in PROBES:
        g["after_step." + k] = named[k].detach().reshape(-1)[:16].numpy().copy()
    g["after_step_bn1_running_mean"] = ref.backbone.visual.bn1.running_mean.numpy().copy()

    lo, grads = O.train_step(sd, auxsd, batch, faithful=False)
    print("G5 losses ref", g["losses"], "oracle", [lo[k] for k in ("loss", "l1", "l4", "l5")])
    worst = 0.0
    for k in PROBES:
        rel = abs(float(grads[k].norm()) - float(g["grad_norm." + k])) / (float(g["grad_norm." + k]) + 1e-20)
        d_after = float(np.abs(sd[k].reshape(-1)[:16].numpy() - g["after_step." + k]).max())
        worst = max(worst, rel)
        print(f"G6 {k:60s} |g| {float(g['grad_norm.' + k]):.4e} rel-dev {rel:.2e} after-step dev {d_after:.2e}")
    print("G6 worst grad-norm rel dev", worst, "; no-grad keys:", len(nograd))
    np.savez_compressed(os.path.join(OUT, "g5_g6_step.npz"), **g)
    seed_fill(ref.state_dict(), 1234)

    # ---- G7 eval post-processing ----------------------------------------------------------
    g = {}
    ref.eval()
    rng = np.random.RandomState(5)
    with torch.no_grad():
        for n, (oh, ow) in enumerate([(427, 640), (480, 333), (321, 500)]):
            o = ref(img[n % 2:n % 2 + 1], ids[n % 2:n % 2 + 1])
            tgt = torch.zeros(oh, ow, dtype=torch.bool)
            y0, x0 = rng.randint(0, oh // 2), rng.randint(0, ow // 2)
            tgt[y0:y0 + oh // 3, x0:x0 + ow // 3] = True
            pred = F.interpolate(o, (oh, ow), align_corners=True, mode="bilinear").squeeze(0)
            pred /= F.adaptive_max_pool2d(pred, (1, 1)) + 1e-5
            pred = pred.squeeze(0)
            cam = pred.clone()
            m = pred.gt(1e-9)
            from utils.util import compute_mask_IU  # reference utils/util.py:9
            I, U = compute_mask_IU(tgt, m)
            oI, oU, om, ocam = O.eval_postprocess(o, tgt)
            am = int(torch.argmax(cam))
            print(f"G7 case {n}: ref I,U = {int(I)},{int(U)}  oracle {oI},{oU}  cam dev {dev(ocam, cam)}")
            g[f"case{n}"] = np.array([oh, ow, y0, x0, int(I), int(U), am])
            g[f"case{n}_cam_ds8"] = cam[::8, ::8].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g7_eval.npz"), **g)

    # ---- G8 tokenizer known answers -------------------------------------------------------
    clip = ref_shim.install()
    sents = ["man on the right", "the woman in a red dress holding an umbrella", "left zebra",
             "a very long sentence that goes on and on and on about the second giraffe from the left side of the picture"]
    toks = clip.tokenize(sents, truncate=True)[:, :20].numpy()
    np.savez_compressed(os.path.join(OUT, "g8_tokenizer.npz"), sentences=np.array(sents), tokens=toks)
    print("G8", toks[0][:8])
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden dir bytes", sz)


if __name__ == "__main__":
    main()
