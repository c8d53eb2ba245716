"""COCO segmentation -> binary mask, the part of `pycocotools.mask` the reference uses (dataset/refer.py:279-291:
`frPyObjects` on polygon lists, `decode`, `area`).

pycocotools (third-party, pinned 2.0.7 in the reference's environment.yml:109) is absent from this image, so its
published algorithm (common/maskApi.c: rleFrPoly / rleDecode / rleFrString) is restated here: a polygon is upsampled
x5, its boundary walked with integer DDA steps, the x-crossings are down-sampled to column-major run boundaries and
sorted into run lengths.  PARITY UNPINNED: there is no pycocotools here to check against; tests pin invariants only
(axis-aligned shapes, area, symmetry).  Used on the host when the dataset cache is built -- not on the GPU path.
"""
import numpy as np


def _poly_to_counts(xy, h, w):
    """Column-major run lengths (starting with a 0-run) of one polygon [x0,y0,x1,y1,...] on an h x w canvas."""
    scale = 5.0
    xy = np.asarray(xy, np.float64)
    k = xy.size // 2
    x = (scale * xy[0:2 * k:2] + 0.5).astype(np.int64)   # C cast of a positive double = truncation
    y = (scale * xy[1:2 * k:2] + 0.5).astype(np.int64)
    x = np.concatenate([x, x[:1]])
    y = np.concatenate([y, y[:1]])
    us, vs = [], []
    for j in range(k):
        xs, xe, ys, ye = int(x[j]), int(x[j + 1]), int(y[j]), int(y[j + 1])
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else 0.0   # 0/0 = nan in C, harmless there because d only takes the value 0
            d = np.arange(dx + 1)
            t = dx - d if flip else d
            us.append(t + xs)
            vs.append((ys + s * t + 0.5).astype(np.int64))
        else:
            s = (xe - xs) / dy
            d = np.arange(dy + 1)
            t = dy - d if flip else d
            vs.append(t + ys)
            us.append((xs + s * t + 0.5).astype(np.int64))
    u = np.concatenate(us)
    v = np.concatenate(vs)
    # points where the boundary crosses a pixel column, down-sampled to pixel units
    ch = np.nonzero(u[1:] != u[:-1])[0] + 1
    uj, up, vj, vp = u[ch], u[ch - 1], v[ch], v[ch - 1]
    xd = np.where(uj < up, uj, uj - 1).astype(np.float64)
    xd = (xd + 0.5) / scale - 0.5
    keep = (np.floor(xd) == xd) & (xd >= 0) & (xd <= w - 1)
    yd = np.where(vj < vp, vj, vp).astype(np.float64)
    yd = (yd + 0.5) / scale - 0.5
    yd = np.ceil(np.clip(yd, 0, h))
    a = (xd[keep].astype(np.int64) * h + yd[keep].astype(np.int64))
    a = np.sort(np.concatenate([a, [h * w]]))
    a = np.diff(np.concatenate([[0], a]))
    # merge zero-length runs (a zero run joins its two neighbours)
    counts = []
    j = 0
    n = a.size
    counts.append(int(a[0]))
    j = 1
    while j < n:
        if a[j] > 0:
            counts.append(int(a[j]))
            j += 1
        else:
            j += 1
            if j < n:
                counts[-1] += int(a[j])
                j += 1
    return counts


def _counts_from_string(s):
    """Compressed RLE string -> counts (maskApi.c rleFrString: 5 bits per char, continuation bit 0x20, delta coding)."""
    if isinstance(s, str):
        s = s.encode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def _decode_counts(counts, h, w):
    m = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            m[pos:pos + c] = 1
        pos += c
        val ^= 1
    return m.reshape((h, w), order="F")


def frPyObjects(seg, h, w):
    """polygon list -> list of RLE dicts; uncompressed RLE dict -> RLE dict"""
    if isinstance(seg, dict):
        return seg
    return [{"size": [h, w], "counts": _poly_to_counts(p, h, w)} for p in seg]


def decode(rle):
    """RLE dict -> uint8 [h,w]; list of RLE dicts -> uint8 [h,w,n]"""
    if isinstance(rle, dict):
        h, w = rle["size"]
        c = rle["counts"]
        return _decode_counts(c if isinstance(c, (list, tuple)) else _counts_from_string(c), h, w)
    return np.stack([decode(r) for r in rle], axis=2)


def area(rle):
    if isinstance(rle, dict):
        c = rle["counts"]
        c = c if isinstance(c, (list, tuple)) else _counts_from_string(c)
        return int(sum(c[1::2]))
    return [area(r) for r in rle]
