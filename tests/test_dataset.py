"""Input pipeline, host side: Pillow-exact resize tables, COCO mask restatement, REFER index /
ReferDataset sampling semantics against the reference-generated fixture g9 and (where mounted) the live reference."""
import hashlib
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import data_oracle as DO
from tris_amd.dataset import cocomask
from tris_amd.dataset.pil_tables import nearest_index, resample_tables
from tris_amd.dataset.ReferDataset import ReferDataset
from tris_amd.dataset.transform import get_transform, normalize_table
from tris_amd.utils.synth import make_mini_refer, word_hash_tokenize

SIZES = [(48, 64), (64, 48), (37, 50), (33, 47), (1, 1), (2, 3), (70, 32), (9, 120), (32, 64), (10, 32), (5, 32), (33, 31)]


@pytest.mark.parametrize("out", [32, 7, 64])
def test_resample_tables_reproduce_pillow_bit_for_bit(out):
    rng = np.random.default_rng(out)
    for h, w in SIZES:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((out, out), Image.BILINEAR))
        assert np.array_equal(DO.pil_resize_bilinear(img, out, out, resample_tables), ref), (h, w, out)
        m = rng.integers(0, 2, (h, w), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(m, mode="P").resize((out, out), Image.NEAREST))
        assert np.array_equal(DO.pil_resize_nearest(m, out, out, nearest_index), ref), (h, w, out)


def test_resample_full_size_case_320():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (427, 640, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((320, 320), Image.BILINEAR))
    assert np.array_equal(DO.pil_resize_bilinear(img, 320, 320, resample_tables), ref)


def test_normalize_table_equals_transform_arithmetic():
    lut = normalize_table()
    assert torch.equal(lut, DO.normalize_lut())
    img = Image.fromarray(np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2))
    x, _ = get_transform(16, train=False)(img, Image.fromarray(np.zeros((16, 16), np.uint8), mode="P"))
    assert torch.equal(x.reshape(3, 256), lut)
    xo, _ = DO.transform(img, Image.fromarray(np.zeros((16, 16), np.uint8), mode="P"), 16, False)
    assert torch.equal(x, xo)


def test_cocomask_invariants():
    h, w = 20, 30
    r = cocomask.frPyObjects([[5, 3, 15, 3, 15, 10, 5, 10]], h, w)
    m = cocomask.decode(r)[:, :, 0]
    ys, xs = np.nonzero(m)
    assert m.sum() == 70 == cocomask.area(r)[0] and (ys.min(), ys.max(), xs.min(), xs.max()) == (3, 9, 5, 14)
    # vertex order / starting vertex do not matter; the mirror image is the mirrored mask
    p = [2.5, 2.5, 20.2, 4.1, 12.7, 17.9]
    a = cocomask.decode(cocomask.frPyObjects([p], h, w))[:, :, 0]
    b = cocomask.decode(cocomask.frPyObjects([p[2:] + p[:2]], h, w))[:, :, 0]
    c = cocomask.decode(cocomask.frPyObjects([[p[4], p[5], p[2], p[3], p[0], p[1]]], h, w))[:, :, 0]
    assert np.array_equal(a, b) and np.array_equal(a, c)
    tri_area = 0.5 * abs((20.2 - 2.5) * (17.9 - 2.5) - (12.7 - 2.5) * (4.1 - 2.5))
    assert abs(int(a.sum()) - tri_area) < 0.08 * tri_area
    # RLE round trip: uncompressed counts and the compressed string form
    rle = {"size": [h, w], "counts": [3, 5, 40, 7, h * w - 55]}
    d = cocomask.decode(rle)
    assert d.sum() == 12 and d.reshape(-1, order="F")[3:8].all()
    assert cocomask._counts_from_string("03") == [0, 3] and cocomask._counts_from_string(b"n0") == [30]
    # empty polygon list member outside the canvas -> empty mask
    assert cocomask.decode(cocomask.frPyObjects([[-9, -9, -5, -9, -5, -5]], h, w)).sum() == 0


def _digest(root):
    h = hashlib.sha256()
    for sub in ("train2014", os.path.join("refer", "refcocog")):
        for f in sorted(os.listdir(os.path.join(root, sub))):
            if f.endswith(".jpg"):
                h.update(np.asarray(Image.open(os.path.join(root, sub, f)).convert("RGB")).tobytes())
            elif f.endswith(".json"):
                h.update(open(os.path.join(root, sub, f), "rb").read())
    return h.hexdigest()


@pytest.fixture(scope="module")
def mini(tmp_path_factory, golden):
    g = golden("g9_dataset.npz")
    n_images, ds_seed, rng_seed, size = (int(v) for v in g["params"])
    root = make_mini_refer(str(tmp_path_factory.mktemp("refer")), n_images=n_images, seed=ds_seed)
    assert _digest(root) == str(g["digest"]), "synthetic dataset differs from the one the fixture was generated on"
    return g, root, rng_seed, size


def test_refer_dataset_matches_reference_fixture(mini):
    g, root, rng_seed, size = mini
    kw = dict(refer_data_root=root, dataset="refcocog", splitBy="umd", size=size, max_tokens=20,
              tokenizer=word_hash_tokenize)
    tr = ReferDataset(image_transforms=get_transform(size, True), split="train", eval_mode=False, negative_samples=3, **kw)
    n = len(tr)
    assert n == g["train_img"].shape[0]
    np.random.seed(rng_seed)
    out = [tr[i] for i in range(n)] + [tr[i] for i in range(n)]
    assert np.array_equal(torch.stack([s["img"] for s, _ in out[:n]]).numpy(), g["train_img"])      # bit-exact pixels
    assert np.array_equal(torch.stack([s["word_ids"] for s, _ in out]).numpy(), g["train_word_ids"])
    assert np.array_equal(torch.stack([s["word_masks"] for s, _ in out]).numpy(), g["train_word_masks"])
    assert np.array_equal(torch.stack([s["neg_word_ids"] for s, _ in out]).numpy(), g["train_neg_word_ids"])
    assert np.array_equal(torch.stack([t["target"] for _, t in out[:n]]).numpy(), g["train_target"])
    assert np.array_equal(np.stack([t["boxes"] for _, t in out[:n]]), g["train_boxes"])
    assert [t["img_path"] for _, t in out[:n]] == list(g["train_img_path"])
    assert np.array_equal(np.stack([t["orig_size"] for _, t in out[:n]]), g["train_orig_size"])
    assert [t["sentences"] for _, t in out] == list(g["train_sentences"])
    assert ["|".join(s["neg_sents"]) for s, _ in out] == list(g["train_neg_sents"])
    ev = ReferDataset(image_transforms=get_transform(size, False), split="val", eval_mode=True, **kw)
    eo = [ev[i] for i in range(len(ev))]
    assert np.array_equal(torch.stack([s["img"] for s, _ in eo]).numpy(), g["val_img"])
    assert [s["word_ids"].shape[-1] for s, _ in eo] == list(g["val_n_sent"])
    assert np.array_equal(torch.cat([s["word_ids"] for s, _ in eo], dim=-1).numpy(), g["val_word_ids"])
    assert [int(t["target"].sum()) for _, t in eo] == list(g["val_target_sum"])
    assert np.array_equal(np.stack([np.array(t["target"].shape) for _, t in eo]), g["val_target_shape"])
    assert np.array_equal(eo[0][1]["target"].numpy(), g["val_target0"])


def test_refer_index_queries(mini):
    from tris_amd.dataset.refer import REFER
    _, root, _, _ = mini
    r = REFER(root, "refcocog", "umd")
    allr = r.getRefIds()
    assert sorted(r.getRefIds(split="train") + r.getRefIds(split="val")) == sorted(allr)
    assert all(r.Refs[i]["split"] == "val" for i in r.getRefIds(split="val"))
    rid = allr[0]
    ref = r.loadRefs(rid)[0]
    assert r.getImgIds(rid) == [ref["image_id"]] and r.getRefBox(rid) == r.Anns[ref["ann_id"]]["bbox"]
    assert set(r.getAnnIds(image_ids=ref["image_id"])) == {a["id"] for a in r.imgToAnns[ref["image_id"]]}
    m = r.getMask(ref)
    img = r.Imgs[ref["image_id"]]
    assert m["mask"].shape == (img["height"], img["width"]) and m["mask"].sum() == m["area"] > 0
    with pytest.raises(SystemExit):
        REFER(root, "nosuchset", "umd")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not mounted on this machine")
def test_refer_dataset_matches_live_reference(tmp_path):
    from oracle import ref_shim
    RefDS, ref_tf = ref_shim.install_dataset()
    root = make_mini_refer(str(tmp_path), n_images=9, seed=23)
    for split, ev, neg in (("train", False, 2), ("val", True, 0), ("train", False, 0)):
        kw = dict(refer_data_root=root, dataset="refcocog", splitBy="umd", split=split, size=40, max_tokens=20,
                  eval_mode=ev, negative_samples=neg)
        a = RefDS(image_transforms=ref_tf(40, train=not ev), **kw)          # real BPE tokenizer on both sides
        b = ReferDataset(image_transforms=get_transform(40, train=not ev), **kw)
        assert len(a) == len(b) and a.ref_ids == b.ref_ids
        np.random.seed(3)
        ra = [a[i] for i in range(len(a))]
        np.random.seed(3)
        rb = [b[i] for i in range(len(b))]
        for (sa, ta), (sb, tb) in zip(ra, rb):
            assert sa.keys() == sb.keys() and ta.keys() == tb.keys()
            for k in sa:
                assert torch.equal(sa[k], sb[k]) and sa[k].dtype == sb[k].dtype if torch.is_tensor(sa[k]) else sa[k] == sb[k], k
            for k in ta:
                if torch.is_tensor(ta[k]):
                    assert torch.equal(ta[k], tb[k]) and ta[k].dtype == tb[k].dtype, k
                elif isinstance(ta[k], np.ndarray):
                    assert np.array_equal(ta[k], tb[k]) and ta[k].dtype == tb[k].dtype, k
                else:
                    assert ta[k] == tb[k], k
