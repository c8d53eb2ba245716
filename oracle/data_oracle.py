"""CPU oracle of the input pipeline.  TEST INFRASTRUCTURE ONLY -- never imported by tris_amd/.

Pinned against Pillow itself (the third-party library that defines the resize arithmetic; present in this image and on
the GPU box) and against the reference's own dataset/transform.py + dataset/ReferDataset.py run through
oracle/ref_shim.py (tests/test_oracle_vs_reference.py, tests/golden/g9_dataset.npz).
"""
import numpy as np
import torch
from PIL import Image

MEAN = (0.485, 0.456, 0.406)   # dataset/transform.py:61
STD = (0.229, 0.224, 0.225)    # dataset/transform.py:62


def resample_pass(img, bounds, kk, axis):
    """One fixed-point pass of Pillow's resampler over `axis` (0 = y, 1 = x) of uint8 [H,W,C] (Resample.c 8bpc loops)."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for o in range(bounds.shape[0]):
        first, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        acc = (1 << 21) + np.tensordot(kk[o, :cnt].astype(np.int64), src[first:first + cnt], axes=(0, 0))
        out[o] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear(img, out_h, out_w, tables):
    """uint8 [H,W,C] -> [out_h,out_w,C]: horizontal pass, uint8 rounding, vertical pass (Resample.c ImagingResampleInner)."""
    h, w = img.shape[:2]
    if w != out_w:
        b, k, _ = tables(w, out_w)
        img = resample_pass(img, b, k, 1)
    if h != out_h:
        b, k, _ = tables(h, out_h)
        img = resample_pass(img, b, k, 0)
    return img


def pil_resize_nearest(img, out_h, out_w, index):
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    yi, xi = index(h, out_h), index(w, out_w)
    out = img[np.clip(yi, 0, None)][:, np.clip(xi, 0, None)]
    out[yi < 0] = 0
    out[:, xi < 0] = 0
    return out


def transform(pil_img, pil_target, size, train):
    """dataset/transform.py:23-63 with torchvision 0.9 semantics on PIL inputs: F.resize -> Image.resize((w,h), BILINEAR)
    (NEAREST for the target, train only), F.to_tensor -> uint8 HWC -> float CHW / 255, torch.tensor(np.asarray(target),
    int64), F.normalize -> (x - mean) / std per channel."""
    img = pil_img.resize((size, size), Image.BILINEAR)
    if train:
        pil_target = pil_target.resize((size, size), Image.NEAREST)
    x = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    t = torch.tensor(np.asarray(pil_target), dtype=torch.int64)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(-1, 1, 1)
    return x.sub_(mean).div_(std), t


def normalize_lut():
    """[3,256] float32: the value `transform` produces for every (channel, byte) -- same float ops, so bit-identical."""
    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255).view(1, 256).repeat(3, 1)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1)
    return v.sub_(mean).div_(std)
