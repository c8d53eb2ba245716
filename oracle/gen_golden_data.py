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


if __name__ == "__main__":
    main()
    main_pixel_attention()
