"""Host-side tables for the bit-exact Pillow resize kernels (`tris_resample_u8`, `tris_gather2d_u8`).

The reference resizes PIL images with torchvision's `F.resize` (dataset/transform.py:29,32; ReferDataset.py:187), i.e.
`Image.resize(..., BILINEAR)` for images and `NEAREST` for masks.  Pillow (third-party, pinned `pillow==10.0.0` in the
reference's environment.yml:103; 12.2 in this image -- same resampler) implements them as

  * BILINEAR: two separable passes (x then y) of a triangle filter whose support is stretched by the down-scale factor
    (antialiasing).  Taps are computed in double, normalised to sum 1, rounded to 22-bit fixed point; each pass
    accumulates `0.5 + sum(pixel * tap)` in int32, shifts and saturates to uint8 (src/libImaging/Resample.c:
    precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc).
  * NEAREST: an affine scale whose source coordinate is *accumulated* (`xo += a0`) in double and truncated
    (src/libImaging/Geometry.c: ImagingScaleAffine).

Only the tables are built here (a few KB per size pair, cached); the pixel work runs on the GPU.
"""
import math
from functools import lru_cache

import numpy as np

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=4096)
def resample_tables(in_size, out_size):
    """-> (bounds int32[out,2], kk int32[out,ksize], ksize) of one axis for the triangle (BILINEAR) filter."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(v * one - 0.5) if v < 0 else int(v * one + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


@lru_cache(maxsize=4096)
def nearest_index(in_size, out_size):
    """-> int32[out] source index per output position (-1 = outside) for Pillow's NEAREST resize."""
    a = float(in_size) / out_size
    o = a * 0.5
    idx = np.empty(out_size, np.int32)
    for x in range(out_size):
        v = -1 if o < 0.0 else int(o)
        idx[x] = v if 0 <= v < in_size else -1
        o += a
    return idx
