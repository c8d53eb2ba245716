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
