"""Generates tests/golden/g9_dataset.npz from the REAL reference dataset code (dataset/ReferDataset.py +
dataset/transform.py imported through oracle/ref_shim.install_dataset) run on the synthetic mini dataset of
tris_amd.utils.synth.make_mini_refer.  TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/gen_golden_data.py

The tokenizer is swapped for tris_amd.utils.synth.word_hash_tokenize on the reference side so that the fixture does
not depend on the BPE vocabulary file (absent on the GPU box); the real tokenizer is pinned by g8_tokenizer.npz.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

N_IMAGES, DS_SEED, RNG_SEED, SIZE = 7, 11, 5, 32


def dataset_digest(root):
    h = hashlib.sha256()
    for sub in ("train2014", os.path.join("refer", "refcocog")):
        for f in sorted(os.listdir(os.path.join(root, sub))):
            if f.endswith(".jpg"):
                from PIL import Image
                h.update(np.asarray(Image.open(os.path.join(root, sub, f)).convert("RGB")).tobytes())
            elif f.endswith(".json"):
                h.update(open(os.path.join(root, sub, f), "rb").read())
    return h.hexdigest()


def main():
    from oracle import ref_shim
    RefDS, ref_tf = ref_shim.install_dataset()
    from tris_amd.utils.synth import make_mini_refer, word_hash_tokenize
    import CLIP.clip as ref_clip
    ref_clip.tokenize = word_hash_tokenize
    root = make_mini_refer(tempfile.mkdtemp(), n_images=N_IMAGES, seed=DS_SEED)
    g = {"digest": np.array(dataset_digest(root)), "params": np.array([N_IMAGES, DS_SEED, RNG_SEED, SIZE])}
    kw = dict(refer_data_root=root, dataset="refcocog", splitBy="umd", size=SIZE, max_tokens=20)
    tr = RefDS(image_transforms=ref_tf(SIZE, train=True), split="train", eval_mode=False, negative_samples=3, **kw)
    np.random.seed(RNG_SEED)
    out = [tr[i] for i in range(len(tr))] + [tr[i] for i in range(len(tr))]
    g["train_img"] = torch.stack([s["img"] for s, _ in out[:len(tr)]]).numpy()
    g["train_word_ids"] = torch.stack([s["word_ids"] for s, _ in out]).numpy()
    g["train_word_masks"] = torch.stack([s["word_masks"] for s, _ in out]).numpy()
    g["train_neg_word_ids"] = torch.stack([s["neg_word_ids"] for s, _ in out]).numpy()
    g["train_target"] = torch.stack([t["target"] for _, t in out[:len(tr)]]).numpy().astype(np.uint8)
    g["train_boxes"] = np.stack([t["boxes"] for _, t in out[:len(tr)]])
    g["train_img_path"] = np.array([t["img_path"] for _, t in out[:len(tr)]])
    g["train_orig_size"] = np.stack([t["orig_size"] for _, t in out[:len(tr)]])
    g["train_sentences"] = np.array([t["sentences"] for _, t in out])
    g["train_neg_sents"] = np.array(["|".join(s["neg_sents"]) for s, _ in out])
    ev = RefDS(image_transforms=ref_tf(SIZE, train=False), split="val", eval_mode=True, **kw)
    eo = [ev[i] for i in range(len(ev))]
    g["val_img"] = torch.stack([s["img"] for s, _ in eo]).numpy()
    g["val_n_sent"] = np.array([s["word_ids"].shape[-1] for s, _ in eo])
    g["val_word_ids"] = torch.cat([s["word_ids"] for s, _ in eo], dim=-1).numpy()
    g["val_target_sum"] = np.array([int(t["target"].sum()) for _, t in eo])
    g["val_target_shape"] = np.stack([np.array(t["target"].shape) for _, t in eo])
    g["val_target0"] = eo[0][1]["target"].numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "g9_dataset.npz"), **g)
    print("g9:", {k: v.shape for k, v in g.items()}, os.path.getsize(os.path.join(OUT, "g9_dataset.npz")))


def pixel_attention_case(seed, N, Ci, Ct, H, W, T):
    """deterministic inputs + weights for the PixelAttention fixture / tests (weights 0.25*randn so the softmax is not flat)"""
    g = torch.Generator().manual_seed(seed)
    shapes = {"Wk.weight": (Ci, Ct, 1), "Wk.bias": (Ci,), "Wv.weight": (Ci, Ct, 1), "Wv.bias": (Ci,)}
    for n in ("Wq", "Wm", "Ww", "Wo"):
        shapes[n + ".weight"] = (Ci, Ci, 1, 1)
        shapes[n + ".bias"] = (Ci,)
    for n in ("ins_q", "ins_w"):
        shapes[n + ".weight"] = (Ci,)
        shapes[n + ".bias"] = (Ci,)
    sd = {}
    for k in sorted(shapes):
        v = torch.randn(shapes[k], generator=g)
        sd[k] = (1.0 + 0.1 * v) if k.startswith("ins_") and k.endswith("weight") else 0.25 * v
    vis = torch.randn(N, Ci, H, W, generator=g)
    lan = torch.randn(N, Ct, T, generator=g)
    return sd, vis, lan


def main_pixel_attention():
    """g10: the REAL reference PixelAttention (model/attn.py:9-65) forward + backward on a small case"""
    from oracle import ref_shim
    ref_shim.install()
    from model.attn import PixelAttention
    N, Ci, Ct, H, W, T = 2, 64, 32, 6, 5, 7
    sd, vis, lan = pixel_attention_case(3, N, Ci, Ct, H, W, T)
    m = PixelAttention(Ci, Ct)
    m.load_state_dict(sd)
    vis.requires_grad_(True)
    lan.requires_grad_(True)
    out = m(vis, lan)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    out.backward(gout)
    g = {"dims": np.array([N, Ci, Ct, H, W, T]), "out": out.detach().numpy(), "gout": gout.numpy(),
         "dvis": vis.grad.numpy(), "dlan": lan.grad.numpy()}
    for k, p in m.named_parameters():
        g["d_" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g10_pixel_attention.npz"), **g)
    print("g10:", {k: v.shape for k, v in g.items()})


VIT_CASE = dict(input_resolution=64, patch_size=16, width=128, layers=2, heads=2, output_dim=64, image=96, batch=2)


def main_vit_spatial():
    """g11: the dense-trunk ViT variant the reference keeps in comments (CLIP/clip/model.py:427-441), executed with the
    REAL reference sub-modules (conv1, ln_pre, transformer) and those commented lines as glue -- small configuration."""
    from oracle import ref_shim
    ref_shim.install()
    import torch.nn.functional as F
    from CLIP.clip.model import VisionTransformer
    from tris_amd.utils.synth import seed_fill
    c = VIT_CASE
    v = VisionTransformer(c["input_resolution"], c["patch_size"], c["width"], c["layers"], c["heads"], c["output_dim"]).float()
    sd = v.state_dict()
    seed_fill(sd, 77)
    with torch.no_grad():   # widen the spread so the softmax is not flat
        for k in sd:
            if sd[k].dim() > 1 or k == "class_embedding":
                sd[k].mul_(8.0)
    v.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    img = torch.randn(c["batch"], 3, c["image"], c["image"], generator=g)
    x = v.conv1(img)
    H, W = x.shape[-2:]
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([v.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype), x], dim=1)
    sdim = c["input_resolution"] // c["patch_size"]
    cls_pos = v.positional_embedding[0:1, :]
    spatial_pos = F.interpolate(v.positional_embedding[1:, ].reshape(1, sdim, sdim, c["width"]).permute(0, 3, 1, 2),
                                size=(H, W), mode="bilinear")
    spatial_pos = spatial_pos.reshape(c["width"], H * W).permute(1, 0)
    x = x + torch.cat([cls_pos, spatial_pos], dim=0)
    x = v.ln_pre(x)
    x = v.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
    cls_feats = x[:, 0, :]
    spa = x[:, 1:, :].permute(0, 2, 1).reshape(x.shape[0], -1, H, W)
    gs = torch.randn(spa.shape, generator=g)
    gc = torch.randn(cls_feats.shape, generator=g)
    ((spa * gs).sum() + (cls_feats * gc).sum()).backward()
    out = {"cls": cls_feats.detach().numpy(), "spa": spa.detach().numpy(), "img": img.numpy(), "gs": gs.numpy(),
           "gc": gc.numpy()}
    keep = ("class_embedding", "positional_embedding", "ln_pre.weight", "ln_pre.bias",
            "transformer.resblocks.0.attn.in_proj_weight", "transformer.resblocks.1.mlp.c_proj.weight")
    for k, p_ in v.named_parameters():
        if k in keep:
            out["d_" + k] = p_.grad.numpy()
    out["d_conv1.weight_norm"] = np.array(float(v.conv1.weight.grad.norm()))
    out["d_conv1.weight_head"] = v.conv1.weight.grad.reshape(-1)[:512].numpy()
    np.savez_compressed(os.path.join(OUT, "g11_vit_spatial.npz"), **out)
    print("g11:", {k: v_.shape for k, v_ in out.items() if not k.startswith("d_transformer")},
          os.path.getsize(os.path.join(OUT, "g11_vit_spatial.npz")))


def vit_case_state_dict():
    """the seed-filled weights of the g11 case as a flat {'visual.<key>': tensor} dict (shapes from VIT_CASE)"""
    from tris_amd.CLIP.clip.model import VisionTransformer
    from tris_amd.utils.synth import seed_fill
    c = VIT_CASE
    v = VisionTransformer(c["input_resolution"], c["patch_size"], c["width"], c["layers"], c["heads"], c["output_dim"])
    sd = {k: torch.empty(t.shape) for k, t in v.state_dict().items()}
    seed_fill(sd, 77)
    for k in sd:
        if sd[k].dim() > 1 or k == "class_embedding":
            sd[k].mul_(8.0)
    return sd


if __name__ == "__main__":
    main()
    main_pixel_attention()
    main_vit_spatial()
