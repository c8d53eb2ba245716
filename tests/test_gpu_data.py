"""Input pipeline on the GPU: every byte the HIP kernels produce is compared with Pillow (the library
that defines the reference's resize) and with the CPU ReferDataset + DataLoader path -- bit-exact, including at the
full 320 px configuration."""
import numpy as np
import pytest
import torch
from PIL import Image
from torch.utils.data import DataLoader

pytestmark = pytest.mark.gpu

SIZES = [(48, 64), (64, 48), (37, 50), (33, 47), (1, 1), (2, 3), (70, 32), (9, 120), (32, 64), (10, 32), (5, 32), (33, 31),
         (427, 640), (640, 480), (500, 375), (320, 320), (321, 640), (1000, 77)]


@pytest.fixture(scope="module")
def ops():
    from tris_amd import ops
    return ops


@pytest.mark.parametrize("out", [320, 32, 224, 7])
def test_resample_u8_equals_pillow(ops, out):
    rng = np.random.default_rng(out)
    for h, w in SIZES:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((out, out), Image.BILINEAR))
        got = ops.resample_u8(torch.from_numpy(img).cuda(), out, out).cpu().numpy()
        assert np.array_equal(got, ref), (h, w, out, int(np.abs(got.astype(int) - ref).max()))
        ref2 = np.asarray(Image.fromarray(img).resize((out + 3, out), Image.BILINEAR))       # non-square target
        got2 = ops.resample_u8(torch.from_numpy(img).cuda(), out, out + 3).cpu().numpy()
        assert np.array_equal(got2, ref2), (h, w, out)


@pytest.mark.parametrize("out", [320, 32, 7])
def test_resize_nearest_u8_equals_pillow(ops, out):
    rng = np.random.default_rng(out + 1)
    for h, w in SIZES:
        m = rng.integers(0, 2, (h, w), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(m, mode="P").resize((out, out), Image.NEAREST))
        got = ops.resize_nearest_u8(torch.from_numpy(m).cuda(), out, out).cpu().numpy()
        assert np.array_equal(got, ref), (h, w, out)


@pytest.mark.parametrize("planar", [False, True])
def test_gather_normalize_equals_transform(ops, planar):
    from oracle import data_oracle as DO
    from tris_amd.dataset.transform import normalize_table
    rng = np.random.default_rng(0)
    S = 320
    cache = rng.integers(0, 256, (5, S, S, 3), dtype=np.uint8)
    idx = torch.tensor([4, 0, 0, 3, 1, 2, 4], dtype=torch.int64)
    got = ops.gather_normalize(torch.from_numpy(cache).cuda(), idx.cuda(), normalize_table().cuda(), planar=planar)
    assert got.shape == (7, 3, S, S)
    assert got.is_contiguous() == planar and got.permute(0, 2, 3, 1).is_contiguous() == (not planar)
    blank = Image.fromarray(np.zeros((S, S), np.uint8), mode="P")
    ref = torch.stack([DO.transform(Image.fromarray(cache[i]), blank, S, False)[0] for i in idx.tolist()])
    assert torch.equal(got.cpu(), ref)          # bit-exact fp32
    # the channels-last batch enters the trunk without a layout pass
    x = ops.nchw_to_nhwc(got)
    assert x.shape == (7, S, S, 3) and torch.equal(x.cpu(), ref.permute(0, 2, 3, 1))
    if not planar:
        assert x.data_ptr() == got.data_ptr()


def test_gather_rows(ops):
    t = torch.arange(7 * 20, dtype=torch.int32).view(7, 20).cuda()
    i = torch.tensor([6, 0, 3, 3], dtype=torch.int64).cuda()
    assert torch.equal(ops.gather_rows(t, i), t[i])
    t8 = torch.randint(0, 255, (5, 64), dtype=torch.uint8).cuda()
    assert torch.equal(ops.gather_rows(t8, i[1:]), t8[i[1:]])
    t64 = torch.arange(40, dtype=torch.int64).view(10, 4).cuda()
    assert torch.equal(ops.gather_rows(t64, i), t64[i])
    assert ops.gather_rows(t64, i[:0]).shape == (0, 4)


def _same(a, b, path=""):
    if torch.is_tensor(a):
        assert torch.is_tensor(b) and a.dtype == b.dtype and a.shape == b.shape, (path, a.dtype, b.dtype, a.shape, b.shape)
        assert torch.equal(a.cpu(), b.cpu()), path
    elif isinstance(a, dict):
        assert a.keys() == b.keys(), (path, a.keys(), b.keys())
        for k in a:
            _same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("size", [32, 320])
def test_hbm_loader_equals_cpu_dataloader(tmp_path, size):
    from tris_amd.dataset.hbm import HbmLoader, HbmReferCache
    from tris_amd.dataset.ReferDataset import ReferDataset
    from tris_amd.dataset.transform import get_transform
    from tris_amd.utils.synth import make_mini_refer, word_hash_tokenize
    root = make_mini_refer(str(tmp_path), n_images=9, seed=4, max_side=72 if size == 32 else 500)
    kw = dict(refer_data_root=root, dataset="refcocog", splitBy="umd", size=size, max_tokens=20, tokenizer=word_hash_tokenize)
    tr = ReferDataset(image_transforms=get_transform(size, True), split="train", eval_mode=False, negative_samples=3, **kw)
    cache = HbmReferCache(tr, size)
    for bs, drop in ((4, False), (5, True)):
        np.random.seed(9)
        ref = list(DataLoader(tr, batch_size=bs, shuffle=False, num_workers=0, drop_last=drop))
        np.random.seed(9)
        got = list(HbmLoader(cache, batch_size=bs, drop_last=drop))
        assert len(got) == len(ref) == len(HbmLoader(cache, batch_size=bs, drop_last=drop))
        for (gs, gt), (rs, rt) in zip(got, ref):
            _same(gs, rs, "samples")
            _same(gt, rt, "targets")
            assert gs["img"].permute(0, 2, 3, 1).is_contiguous()
    ev = ReferDataset(image_transforms=get_transform(size, False), split="val", eval_mode=True, **kw)
    ecache = HbmReferCache(ev, size)
    ref = list(DataLoader(ev, batch_size=1, shuffle=False, num_workers=0))
    got = list(HbmLoader(ecache, batch_size=1))
    assert len(ref) == len(got) > 0
    for (gs, gt), (rs, rt) in zip(got, ref):
        _same(gs, rs, "samples")
        _same(gt, rt, "targets")
    # a sampler (here: a DistributedSampler-like index list) drives the order
    np.random.seed(1)
    a = list(HbmLoader(cache, batch_size=2, sampler=[5, 1, 0, 7]))
    np.random.seed(1)
    b = [default for default in DataLoader(torch.utils.data.Subset(tr, [5, 1, 0, 7]), batch_size=2)]
    for (gs, gt), (rs, rt) in zip(a, b):
        _same(gs, rs, "samples")
        _same(gt, rt, "targets")


def test_train_step_from_hbm_batch_equals_planar_batch(tmp_path):
    """The channels-last batch from the loader gives the same Stage-1 forward as the same pixels in NCHW."""
    from tris_amd.dataset.hbm import HbmLoader, HbmReferCache
    from tris_amd.dataset.ReferDataset import ReferDataset
    from tris_amd.dataset.transform import get_transform
    from tris_amd.args import get_parser
    from tris_amd.model.model_stage1 import TRIS
    from tris_amd.utils.synth import make_mini_refer, seed_fill, word_hash_tokenize
    root = make_mini_refer(str(tmp_path), n_images=6, seed=2)
    tr = ReferDataset(root, "refcocog", "umd", image_transforms=get_transform(64, True), split="train", eval_mode=False,
                      size=64, max_tokens=20, negative_samples=3, tokenizer=word_hash_tokenize)
    cache = HbmReferCache(tr, 64)
    np.random.seed(0)
    s_cl, _ = next(iter(HbmLoader(cache, batch_size=2)))
    np.random.seed(0)
    s_pl, _ = next(iter(HbmLoader(cache, batch_size=2, planar=True)))
    assert torch.equal(s_cl["img"], s_pl["img"])
    args = get_parser().parse_args(["--backbone", "clip-RN50", "--size", "64", "--max_query_len", "20"])
    model = TRIS(args).cuda()
    sd = model.state_dict()
    seed_fill(sd, 1234)
    model.load_state_dict(sd)
    model.train()
    ids = s_cl["word_ids"].squeeze(1).long()
    with torch.no_grad():
        a = model(s_cl["img"], ids)
        b = model(s_pl["img"], ids)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_script_level_main_trains_validates_checkpoints_and_resumes(tmp_path):
    """python -m tris_amd.train_stage1 equivalent on the synthetic dataset: HBM loader -> train_one_epoch -> validate ->
    best checkpoint; then --resume --eval reproduces the stored validation numbers (train_stage1.py:44-262)."""
    import os
    from tris_amd.args import get_parser
    from tris_amd.train_stage1 import main, setup_seed
    from tris_amd.utils.synth import make_mini_refer, word_hash_tokenize
    root = make_mini_refer(str(tmp_path / "data"), n_images=8, seed=6)
    out = str(tmp_path / "out")
    argv = ["--refer_data_root", root, "--dataset", "refcocog", "--splitBy", "umd", "--size", "64", "--batch_size", "4",
            "--epoch", "2", "--test_split", "val", "--output", out, "--negative_samples", "3", "--print-freq", "1",
            "--backbone", "clip-RN50", "--max_query_len", "20"]
    setup_seed(1234)
    best = main(get_parser().parse_args(argv), tokenizer=word_hash_tokenize)
    assert best["epoch"] >= 0 and os.path.exists(best["path"]) and os.path.exists(best["hit_path"])
    ck = torch.load(best["path"], map_location="cpu")
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch"} and len(ck["model"]) == 518
    res = main(get_parser().parse_args(argv + ["--resume", "--eval", "--pretrain", os.path.basename(best["path"])]),
               tokenizer=word_hash_tokenize)
    oIoU, mIoU, hit = res[0]
    assert abs(float(mIoU) - best["val_acc"]) < 1e-3 and hit == best["val_hit"]
    # the CPU DataLoader path (the reference's loader) feeds the same loop
    from tris_amd.config import cfg
    with cfg.override(hbm_loader=False):
        res2 = main(get_parser().parse_args(argv + ["--resume", "--eval", "--pretrain", os.path.basename(best["path"])]),
                    tokenizer=word_hash_tokenize)
    assert abs(float(res2[0][1]) - float(mIoU)) < 1e-4 and res2[0][2] == hit
