"""Deterministic weights and synthetic batches.

There are no CLIP weights and no RefCOCO data on either machine, so every parity test,
the benchmark and the CPU baseline share:

* `seed_fill(state_dict, seed)` -- fill a state dict (reference *or* ours: same keys, same
  shapes) deterministically.  The stock random init is degenerate (bn3.weight = 0,
  CLIP/clip/model.py:520-523 in the reference), so a seed-fill is needed to get non-trivial
  response maps.
* `synthetic_batch(B, size, L, negatives, seed)` -- images ~ N(0,1), token ids shaped like
  CLIP tokeniser output: SOT 49406, 2..17 random ids in [1, 49405], EOT 49407, zero padding.
"""
import numpy as np
import torch

SOT, EOT = 49406, 49407


def seed_fill(sd, seed=1234):
    """In-place deterministic fill, iterating sorted keys with ONE generator.

    Rules (by key suffix / rank):  num_batches_tracked -> 0;  logit_scale kept;
    running_mean 0.1*randn;  running_var 1+0.1*rand;  1-D `.weight` (BN/LN/IN scale)
    1+0.1*randn;  everything else (conv/linear/embedding/proj weights, all biases) 0.02*randn.
    Values are generated in fp32 on CPU and copied, so the result does not depend on the
    device or memory format of the destination tensors.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    for k in sorted(sd.keys()):
        t = sd[k]
        if k.endswith("num_batches_tracked"):
            t.zero_()
            continue
        if k.endswith("logit_scale"):
            continue
        shape = tuple(t.shape)
        if k.endswith("running_mean"):
            v = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith("running_var"):
            v = 1.0 + 0.1 * torch.rand(shape, generator=g)
        elif t.dim() == 1 and k.endswith(".weight"):
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            v = 0.02 * torch.randn(shape, generator=g)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return sd


def heavy_tail_fill(sd, seed=1234, span=2.0, outlier=16.0, frac=0.01):
    """seed_fill, then statistics that look like released CLIP weights rather than iid noise: every weight MATRIX gets a
    log-uniform gain 2^U(-span, span) per output channel, `frac` of its channels a further factor `outlier`; every norm scale
    (1-D `.weight`) a gain 2^U(-span/2, span/2) per channel, `frac` of them x sqrt(outlier).  Deterministic (one generator,
    sorted keys, after the plain fill).  Used by the parity tests to stress per-tensor operand scaling (heavy tails, outlier
    channels); the oracle receives the SAME state dict."""
    seed_fill(sd, seed)
    g = torch.Generator(device="cpu").manual_seed(seed + 977)
    for k in sorted(sd.keys()):
        t = sd[k]
        if not t.is_floating_point() or k.endswith(("running_mean", "running_var", "logit_scale")) or t.dim() == 0:
            continue
        n = t.shape[0]
        if t.dim() == 1 and k.endswith(".weight"):
            gain = torch.exp2((torch.rand(n, generator=g) * 2 - 1) * (span / 2))
            gain[torch.rand(n, generator=g) < frac] *= outlier ** 0.5
        elif t.dim() >= 2 and k.endswith(("weight", "in_proj_weight", "text_projection", "proj")):
            gain = torch.exp2((torch.rand(n, generator=g) * 2 - 1) * span)
            gain[torch.rand(n, generator=g) < frac] *= outlier
        else:
            continue
        with torch.no_grad():
            t.mul_(gain.view(-1, *([1] * (t.dim() - 1))).to(t.device, t.dtype))
    return sd


def synthetic_ids(n, L=20, rng=None):
    """`n` token rows of length L; EOT is the arg-max id so the pooled token is well defined."""
    rng = rng if rng is not None else np.random.RandomState(0)
    ids = np.zeros((n, L), dtype=np.int64)
    for i in range(n):
        m = int(rng.randint(2, min(17, L - 2) + 1))
        ids[i, 0] = SOT
        ids[i, 1:1 + m] = rng.randint(1, 49406, size=m)
        ids[i, 1 + m] = EOT
    return ids


def synthetic_batch(B, size=320, L=20, negatives=3, seed=7, rank=0):
    """Returns dict(img float32 [B,3,size,size], word_ids int64 [B,L], neg_word_ids [B,neg,L])."""
    g = torch.Generator(device="cpu").manual_seed(seed + rank)
    img = torch.randn(B, 3, size, size, generator=g)
    rng = np.random.RandomState(0 + rank)
    word_ids = torch.from_numpy(synthetic_ids(B, L, rng))
    out = {"img": img, "word_ids": word_ids}
    if negatives > 0:
        out["neg_word_ids"] = torch.from_numpy(synthetic_ids(B * negatives, L, rng)).view(B, negatives, L)
    return out


# ---- synthetic RefCOCO-style dataset on disk (input-pipeline tests / bench) --------------------------
_WORDS = ("the left right red blue small large cup dog person table chair near behind front of man woman holding "
          "white black top bottom second third giraffe plate car sitting standing").split()


def make_mini_refer(root, n_images=6, seed=0, dataset="refcocog", splitBy="umd", sizes=None, max_side=72):
    """Write a tiny dataset in the layout dataset/refer.py:46-78 reads:
    <root>/refer/<dataset>/refs(<splitBy>).p, instances.json, <root>/train2014/COCO_train2014_<id>.jpg
    (files hold losslessly PNG-encoded pixels so decoding is bit-stable).  Image k has 1..3 annotated polygons, each a
    ref with 1..3 sentences; every third image has a single ref (exercises the dataset-wide negative draw) and one ref
    in four belongs to the 'val' split (exercises the split filter and the same-image scan)."""
    import json
    import os
    import pickle
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "refer", dataset), exist_ok=True)
    os.makedirs(os.path.join(root, "train2014"), exist_ok=True)
    images, anns, refs = [], [], []
    ann_id, ref_id, sent_id = 1000, 0, 0
    for k in range(n_images):
        if sizes is not None:
            h, w = sizes[k % len(sizes)]
        else:
            h, w = int(rng.randint(24, max_side)), int(rng.randint(24, max_side))
        coarse = rng.randint(0, 256, (max(2, h // 8), max(2, w // 8), 3)).astype(np.uint8)
        pix = np.asarray(Image.fromarray(coarse).resize((w, h), Image.BICUBIC)).copy()
        pix ^= rng.randint(0, 8, pix.shape).astype(np.uint8)
        img_id = 1 + 7 * k
        fname = "COCO_train2014_%012d.jpg" % img_id
        Image.fromarray(pix).save(os.path.join(root, "train2014", fname), format="PNG")
        images.append({"id": img_id, "file_name": fname, "height": h, "width": w})
        n_obj = 1 if k % 3 == 2 else int(rng.randint(2, 4))
        for _ in range(n_obj):
            cx, cy = rng.uniform(0.25, 0.75) * w, rng.uniform(0.25, 0.75) * h
            rad = rng.uniform(0.1, 0.24) * min(h, w)
            nv = int(rng.randint(3, 8))
            ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
            poly = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1).round(2)
            x0, y0 = poly.min(0)
            x1, y1 = poly.max(0)
            cat = int(rng.randint(1, 4))
            anns.append({"id": ann_id, "image_id": img_id, "category_id": cat, "iscrowd": 0,
                         "segmentation": [poly.reshape(-1).tolist()],
                         "bbox": [float(x0), float(y0), float(x1 - x0), float(y1 - y0)],
                         "area": float((x1 - x0) * (y1 - y0))})
            sents = []
            for _s in range(int(rng.randint(1, 4))):
                words = [str(_WORDS[i]) for i in rng.randint(0, len(_WORDS), int(rng.randint(2, 9)))]
                sents.append({"sent_id": sent_id, "sent": " ".join(words), "raw": " ".join(words), "tokens": words})
                sent_id += 1
            refs.append({"ref_id": ref_id, "ann_id": ann_id, "image_id": img_id, "category_id": cat,
                         "split": "val" if ref_id % 4 == 3 else "train", "file_name": fname,
                         "sent_ids": [s["sent_id"] for s in sents], "sentences": sents})
            ann_id += 1
            ref_id += 1
    with open(os.path.join(root, "refer", dataset, "refs(%s).p" % splitBy), "wb") as f:
        pickle.dump(refs, f)
    with open(os.path.join(root, "refer", dataset, "instances.json"), "w") as f:
        json.dump({"images": images, "annotations": anns,
                   "categories": [{"id": i, "name": "cat%d" % i} for i in (1, 2, 3)]}, f)
    return root


def word_hash_tokenize(texts, context_length=77, truncate=False):
    """Stand-in tokenizer with `clip.tokenize`'s signature and output format (int32 [n, context_length], SOT ... EOT,
    zero padded) for machines without the BPE vocabulary file: one id in [1000, 40000) per whitespace word."""
    import zlib
    import torch
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.int)
    for i, t in enumerate(texts):
        ids = [49406] + [1000 + zlib.crc32(w.encode()) % 39000 for w in t.lower().split()] + [49407]
        ids = ids[:context_length]
        out[i, :len(ids)] = torch.tensor(ids)
    return out
