"""Host-side logic that needs no GPU: drop-in surface (constructor, state-dict keys, argparse defaults, tokenizer),
synthetic inputs, and the evaluation helpers."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_args_defaults_match_reference_recipe():
    from tris_amd.args import get_parser
    a = get_parser().parse_args([])
    assert (a.backbone, a.hidden_dim, a.max_query_len, a.negative_samples) == ("clip-RN50", 1024, 20, 0)
    assert (a.lr, a.lr_multi, a.weight_decay, a.attn_multi) == (5e-5, 0.1, 0.01, 0.1)
    assert (a.w1, a.w4, a.w5, a.FOCAL_P, a.FOCAL_LAMBDA) == (1, 5, 2, 3, 0.01)
    a = get_parser().parse_args(["--weight_decay", "0.1", "--print-freq", "7", "--distributed"])
    assert a.weight_decay == 0.1 and a.print_freq == 7 and a.distributed


def test_tris_module_surface():
    from tris_amd.utils.shapes import _build_tris
    m = _build_tris()
    bb, new = m.trainable_parameters()
    ids = {id(p) for p in bb + new}
    assert id(m.logit_scale) not in ids  # never optimised (model_stage1.py:44-52)
    assert len(bb) == 324 and len(new) == 28
    nograd = [k for k, p in m.named_parameters() if getattr(p, "_tris_no_grad_path", False)]
    assert len(nograd) == 10 and all("attnpool" in k or k == "backbone.logit_scale" for k in nograd)
    w = m.backbone.visual.layer1[0].conv2.weight
    assert w.shape == (64, 64, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    # a reference-layout (contiguous NCHW) checkpoint loads into the channels_last parameters unchanged in value
    sd = {k: v.clone().contiguous() for k, v in m.state_dict().items()}
    sd["backbone.visual.layer1.0.conv2.weight"] = torch.arange(64 * 64 * 9, dtype=torch.float32).view(64, 64, 3, 3)
    m.load_state_dict(sd)
    assert m.backbone.visual.layer1[0].conv2.weight.is_contiguous(memory_format=torch.channels_last)
    assert float(m.backbone.visual.layer1[0].conv2.weight[3, 5, 1, 2]) == float(sd["backbone.visual.layer1.0.conv2.weight"][3, 5, 1, 2])
    with pytest.raises(ValueError):
        _build_tris(["--backbone", "clip-RN50x4"])  # no Stage-1 definition in the reference either
    v = _build_tris(["--backbone", "clip-ViT-B/16"])  # BASELINE configs[4]: the dense ViT trunk (self-defined)
    assert v.vit_trunk and v.vis_project.weight.shape[:2] == (1024, 768) and v.lan_project.weight.shape == (1024, 512)
    assert v.backbone.visual.positional_embedding.shape == (14 * 14 + 1, 768)


def test_tokenizer_known_answers():
    from tris_amd.CLIP.clip import simple_tokenizer
    try:
        simple_tokenizer.default_bpe()
    except FileNotFoundError:
        pytest.skip("CLIP BPE merge table not available on this machine")
    from tris_amd.CLIP import clip
    g = np.load(os.path.join(GOLDEN, "g8_tokenizer.npz"))
    toks = clip.tokenize(list(g["sentences"]), truncate=True)[:, :20].numpy()
    assert (toks == g["tokens"]).all()
    assert clip.tokenize("man on the right")[0, :6].tolist() == [49406, 786, 525, 518, 1155, 49407]
    with pytest.raises(RuntimeError):
        clip.tokenize("word " * 100)


def test_synthetic_batch_is_deterministic_and_rank_sharded():
    from tris_amd.utils.synth import synthetic_batch
    a, b = synthetic_batch(3, 32, 20, 3, seed=7), synthetic_batch(3, 32, 20, 3, seed=7)
    assert torch.equal(a["img"], b["img"]) and torch.equal(a["neg_word_ids"], b["neg_word_ids"])
    c = synthetic_batch(3, 32, 20, 3, seed=7, rank=1)
    assert not torch.equal(a["img"], c["img"]) and not torch.equal(a["word_ids"], c["word_ids"])
    ids = a["word_ids"]
    assert (ids[:, 0] == 49406).all() and (ids.max(1).values == 49407).all() and ids.shape == (3, 20)
    assert a["neg_word_ids"].shape == (3, 3, 20)


def test_eval_helpers():
    from tris_amd.utils.util import AverageMeter, compute_mask_IU
    from tris_amd.validate import isCorrectHit
    m = torch.zeros(4, 6, dtype=torch.bool)
    t = torch.zeros(4, 6, dtype=torch.bool)
    m[:2] = True
    t[1:3] = True
    I, U = compute_mask_IU(m, t)
    assert (int(I), int(U)) == (6, 18)
    with pytest.raises(ValueError):
        compute_mask_IU(m, t[:, :3])
    am = AverageMeter()
    am.update(2.0, 2)
    am.update(4.0, 2)
    assert am.avg == 3.0
    heat = np.zeros((5, 5), dtype=np.float32)
    heat[3, 1] = 2.0
    gt = np.zeros((5, 5), dtype=bool)
    gt[3, 1] = True
    assert isCorrectHit([[0, 2, 2, 4]], heat, gt) == (1, (3, 1), 1)
    assert isCorrectHit([[3, 3, 4, 4]], heat, gt)[0] == 0


def test_checkpoint_roundtrip(tmp_path):
    from types import SimpleNamespace
    from tris_amd.utils.util import load_checkpoint, save_checkpoint
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: 1.0)
    args = SimpleNamespace(output=str(tmp_path), pretrain="c.pth", eval=False, start_epoch=0)
    save_checkpoint(4, net, opt, sched, args=args, checkpoint_name="c.pth")
    ck = torch.load(tmp_path / "c.pth")
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch"}
    net2 = torch.nn.Linear(3, 2)
    load_checkpoint(args, net2, torch.optim.AdamW(net2.parameters()), None)
    assert args.start_epoch == 5 and torch.equal(net2.weight, net.weight)


def test_gradient_segments_cover_the_arenas_in_backward_order():
    """GradReducer.plan with the product's rule set on the real parameter list (arena layout emulated on CPU): every
    arena slot belongs to exactly one segment, the trunk stages map to contiguous ranges, and the text encoder is NOT
    in the segment that is released behind layer4 (tests/test_ddp_order.py runs the graph)."""
    from types import SimpleNamespace
    from tris_amd.parallel import STAGE1_RULES, GradReducer
    from tris_amd.utils.shapes import _build_tris
    m = _build_tris()
    bb, new = m.trainable_parameters()
    arenas = []
    for group in (bb, new):
        ps = [p for p in group if not getattr(p, "_tris_no_grad_path", False)]
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 63) // 64 * 64
        arenas.append(SimpleNamespace(params=ps, offsets=offs, numel=n))
    seg, par = GradReducer.plan(arenas, list(m.named_parameters()), STAGE1_RULES, with_params=True)
    cover = [0, 0]
    for k, ranges in seg.items():
        for ai, s, e in ranges:
            assert e > s
            cover[ai] += e - s
    assert cover == [arenas[0].numel, arenas[1].numel]
    assert len(seg["layer4"]) == 1 and len(seg["layer1"]) == 1 and len(seg["stem"]) == 1
    assert seg["heads"] == [(1, 0, arenas[1].numel)]            # the whole second arena, nothing of the backbone
    assert all(ai == 0 for k in ("text", "text_mid", "text_hi") for ai, *_ in seg[k]) and 1 <= len(seg["text"]) <= 4   # text_projection | transformer | ln_final
    for ai, s, e, n in (r for ranges in par.values() for r in ranges):
        assert 0 <= s < e <= arenas[ai].numel


def test_clip_load_refuses_to_train_from_random_weights(monkeypatch, tmp_path):
    """no weights file and no explicit opt-in -> clip.load raises (the reference downloads or fails, clip.py:43-72)"""
    import pytest
    from tris_amd.CLIP import clip
    monkeypatch.delenv("TRIS_RANDOM_INIT", raising=False)
    clip.allow_random_init(False)
    try:
        with pytest.raises(FileNotFoundError):
            clip.load("ViT-B/32", device="cpu", download_root=str(tmp_path), txt_length=20)
        with clip.random_init():
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m, _ = clip.load("ViT-B/32", device="cpu", download_root=str(tmp_path), txt_length=20)
            assert m.txt_length == 20
        with pytest.raises(FileNotFoundError):
            clip.load("ViT-B/32", device="cpu", download_root=str(tmp_path), txt_length=20)
    finally:
        clip._RANDOM_INIT_OK = None


def test_eval_shard_sampler_does_not_pad():
    """evaluation shards partition the dataset exactly (DistributedSampler would repeat refs to even them out, and the
    all-reduced I/U accumulators of tris_amd.validate would count those twice)"""
    from tris_amd.parallel import ShardSampler
    ds = list(range(10))
    shards = [list(ShardSampler(ds, rank=r, world=4)) for r in range(4)]
    assert sorted(i for s in shards for i in s) == ds
    assert [len(ShardSampler(ds, rank=r, world=4)) for r in range(4)] == [3, 3, 2, 2]


def test_batchnorm_step_counter_is_flushed_when_observed():
    """num_batches_tracked is counted on the host during training and materialised in state_dict / reset by load"""
    from tris_amd.CLIP.clip.model import BatchNorm2d
    m = BatchNorm2d(8)
    m._nbt_pending = 3                      # what three training forwards leave behind
    sd = m.state_dict()
    assert int(sd["num_batches_tracked"]) == 3 and m._nbt_pending == 0
    m._nbt_pending = 2
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    assert int(m.num_batches_tracked) == 3 and m._nbt_pending == 0


def test_gradient_box_protocol():
    """ops.GradBox hands ONE gradient from a later layer's backward to an earlier layer's kernel epilogue: the first deposit
    is taken over, a second one (or one arriving after the consumer has run) goes back to autograd."""
    from tris_amd.ops import GradBox
    box = GradBox()
    g1, g2 = torch.ones(3), torch.zeros(3)
    assert box.value is None and not box.consumed
    assert box.deposit(g1) and box.value is g1
    assert not box.deposit(g2) and box.value is g1          # occupied: the second gradient stays with autograd
    box.value, box.consumed = None, True                     # what the consuming backward does
    assert not box.deposit(g2) and box.value is None         # too late: the consumer has already run


def test_cross_attention_pixel_row_plan_covers_every_pixel_and_channel_unit_once():
    """csrc/xattn_px.hip cuts an image into S pixel ranges and S ranges of 32-channel units by integer division; the host-side
    plan (tris_xattn_px_slots, no GPU needed) must respect the kernel's limits -- B * S workgroups resident one per CU, <= 32
    own pixels (two MFMA tiles), <= 8 own units (one per wave) -- and the ranges must tile [0, P) and [0, U) exactly."""
    from tris_amd import _lib
    lib = _lib.load()
    seen_s = set()
    for cus in (256, 304, 64):
        for C in (512, 1024):
            U = C // 32
            for B in (1, 2, 3, 7, 16, 32, 40, 48, 51, 60, 64, 65, 100, 128, 300):
                for P in (1, 5, 8, 31, 32, 33, 57, 97, 100, 103, 104):
                    S = int(lib.tris_xattn_px_slots(B, P, C, cus))
                    if S == 0:
                        # declines only when no S in 1..8 satisfies all three limits
                        assert not any(B * s <= cus and s <= P and -(-P // s) <= 32 and -(-U // s) <= 8 for s in range(1, 9)), (B, P, C, cus)
                        continue
                    seen_s.add(S)
                    assert 1 <= S <= 8 and B * S <= cus and S <= P
                    px = [((s * P) // S, ((s + 1) * P) // S) for s in range(S)]
                    un = [((s * U) // S, ((s + 1) * U) // S) for s in range(S)]
                    assert px[0][0] == 0 and px[-1][1] == P and all(a[1] == b[0] for a, b in zip(px, px[1:]))
                    assert un[0][0] == 0 and un[-1][1] == U and all(a[1] == b[0] for a, b in zip(un, un[1:]))
                    assert all(1 <= hi - lo <= 32 for lo, hi in px) and all(1 <= hi - lo <= 8 for lo, hi in un)
    assert seen_s >= {4, 5, 6, 8}
    assert lib.tris_xattn_px_slots(48, 100, 1024, 256) == 5      # the headline shape: 240 workgroups on 256 CUs
    assert lib.tris_xattn_px_slots(48, 105, 1024, 256) == 0 and lib.tris_xattn_px_slots(48, 100, 768, 256) == 0
