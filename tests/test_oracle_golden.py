"""Pins oracle/tris_oracle.py against vectors generated from the real reference
(oracle/gen_golden.py).  CPU only; runs anywhere (no /root/reference needed)."""
import numpy as np
import pytest
import torch

from oracle import tris_oracle as O
from tris_amd.utils.synth import seed_fill, synthetic_batch
from tris_amd.utils.shapes import tris_state_dict_spec, aux_state_dict_spec, empty_state_dict

TOL = 5e-5  # oracle vs reference, fp32 CPU, same op semantics (observed <= 1e-5)


@pytest.fixture(scope="module")
def sd():
    return seed_fill(empty_state_dict(tris_state_dict_spec()), 1234)


@pytest.fixture(scope="module")
def aux():
    return seed_fill(empty_state_dict(aux_state_dict_spec()), 4321)


@pytest.fixture(scope="module")
def batch():
    return synthetic_batch(2, 320, 20, 3, seed=7)


def clone(d):
    return {k: v.clone() for k, v in d.items()}


def test_spec_matches_reference_keys():
    import os
    from conftest import GOLDEN
    lines = open(os.path.join(GOLDEN, "state_dict_keys.txt")).read().strip().splitlines()
    ref = [(l.split(" ", 1)[0], eval(l.split(" ", 1)[1])) for l in lines]
    ours = [(k, list(s)) for k, s, _ in tris_state_dict_spec()]
    assert ours == ref
    assert len(ref) == 518
    lines = open(os.path.join(GOLDEN, "aux_state_dict_keys.txt")).read().strip().splitlines()
    ref = [(l.split(" ", 1)[0], eval(l.split(" ", 1)[1])) for l in lines]
    assert [(k, list(s)) for k, s, _ in aux_state_dict_spec()] == ref


def test_g1_text(sd, batch, golden):
    g = golden("g1_g2_encoders.npz")
    with torch.no_grad():
        x, hidden = O.encode_text(sd, "backbone.", batch["word_ids"])
    assert np.abs(hidden.numpy() - g["text_hidden"]).max() < TOL
    assert abs(float(x.sum()) - g["text_x_sum"][0]) < 1e-2


def test_tokens_behind_eot_cannot_reach_hidden(aux, batch):
    """What a packed text pass would rest on, stated against the oracle's restatement of CLIP.encode_text
    (reference CLIP/clip/model.py:537-564): `hidden` is the row at argmax(ids) = the EOT token, the mask is causal, every other op
    is row-wise -- so nothing at a position behind EOT can reach it.  Rewriting those positions with arbitrary token ids (below the
    EOT id, so the argmax stays) leaves `hidden` BIT-identical; 42 % of the synthetic batches' rows are such positions."""
    ids = torch.cat([batch["word_ids"], batch["neg_word_ids"].reshape(-1, batch["word_ids"].shape[1])], 0).long()
    eot = ids.argmax(-1)
    assert int(eot.min()) >= 1 and int(eot.max()) <= ids.shape[1] - 1
    behind = torch.arange(ids.shape[1])[None, :] > eot[:, None]
    assert behind.float().mean() > 0.2
    junk = ids.clone()
    g = torch.Generator().manual_seed(3)
    junk[behind] = torch.randint(1, 40000, (int(behind.sum()),), generator=g)
    assert torch.equal(junk.argmax(-1), eot)
    with torch.no_grad():
        _, h0 = O.encode_text(aux, "", ids)
        x1, h1 = O.encode_text(aux, "", junk)
    assert torch.equal(h0, h1)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_g2_image(sd, batch, golden, mode):
    g = golden("g1_g2_encoders.npz")
    s = clone(sd)
    with torch.no_grad():
        c = O.encode_image_rn(s, "backbone.", batch["img"], mode == "train")
    for i in range(4):
        assert np.abs(c[i][:, :8, :8, :8].numpy() - g[f"c{i + 1}_{mode}_crop"]).max() < TOL
    if mode == "train":
        assert np.abs(s["backbone.visual.bn1.running_mean"].numpy() - g["bn1_running_mean_after"]).max() < 1e-6


def test_g3_bilateral(sd, golden):
    g = golden("g3_bilateral_prompt.npz")
    gen = torch.Generator().manual_seed(11)
    for B in (1, 2, 4):
        vis = torch.randn(B, 1024, 10, 10, generator=gen)
        vis = vis / vis.norm(dim=1, keepdim=True)
        lan = torch.randn(B, 1024, B, generator=gen)
        lan = lan / lan.norm(dim=1, keepdim=True)
        with torch.no_grad():
            nv, nl = O.bilateral_prompt(sd, "attn_fusion", vis, lan)
        assert np.abs(nl.numpy() - g[f"B{B}_new_lan"]).max() < TOL
        assert np.abs(nv[:, :64].numpy() - g[f"B{B}_new_vis_crop"]).max() < TOL


def test_g4_forward(sd, batch, golden):
    g = golden("g4_tris_forward.npz")
    img, ids = batch["img"], batch["word_ids"]
    with torch.no_grad():
        for B in (1, 2):
            o = O.tris_forward(clone(sd), img[:B], ids[:B], False)
            assert np.abs(o[:, :, ::4, ::4].numpy() - g[f"eval_B{B}_full_ds4"]).max() < TOL
        cls, fg, r, s, ls = O.tris_forward(clone(sd), img, ids, True)
    assert np.abs(cls.numpy() - g["train_cls_out"]).max() < TOL
    assert np.abs(fg.numpy() - g["train_cls_fg"]).max() < TOL
    assert np.abs(r[:, :, ::4, ::4].numpy() - g["train_relu_ds4"]).max() < TOL
    assert np.abs(s[:, :, ::4, ::4].numpy() - g["train_sig_ds4"]).max() < TOL


def test_g5_g6_step(sd, aux, batch, golden):
    g = golden("g5_g6_step.npz")
    s = clone(sd)
    lo, grads = O.train_step(s, clone(aux), batch, faithful=False)
    ref = g["losses"]
    assert abs(lo["loss"] - ref[0]) < 1e-3 and abs(lo["l1"] - ref[1]) < 1e-3
    assert abs(lo["l4"] - ref[2]) < 1e-4 and abs(lo["l5"] - ref[3]) < 1e-4
    for k in [n[len("grad_norm."):] for n in g.files if n.startswith("grad_norm.")]:
        gn = float(g["grad_norm." + k])
        assert abs(float(grads[k].norm()) - gn) <= 2e-4 * gn + 1e-7, k
        head = grads[k].reshape(-1)[:16].numpy()
        assert np.abs(head - g["grad_head." + k]).max() <= 2e-4 * max(1e-3, np.abs(g["grad_head." + k]).max()) + 1e-6, k
        assert np.abs(s[k].reshape(-1)[:16].numpy() - g["after_step." + k]).max() < 1e-6, k
    nograd = set(g["nograd_keys"].tolist())
    assert all(k.startswith("backbone.visual.attnpool") or k == "backbone.logit_scale" for k in nograd)
    assert nograd.isdisjoint(grads.keys())


def test_g5_faithful_equals_lean(sd, aux):
    b = synthetic_batch(2, 320, 20, 3, seed=9)
    with torch.no_grad():
        a = O.stage1_losses(clone(sd), aux, b, faithful=True)
        c = O.stage1_losses(clone(sd), aux, b, faithful=False)
    for k in ("loss", "l1", "l4", "l5"):
        assert abs(float(a[k]) - float(c[k])) < 1e-5


def test_g7_eval(sd, batch, golden):
    g = golden("g7_eval.npz")
    img, ids = batch["img"], batch["word_ids"]
    with torch.no_grad():
        for n in range(3):
            oh, ow, y0, x0, I, U, am = [int(v) for v in g[f"case{n}"]]
            o = O.tris_forward(clone(sd), img[n % 2:n % 2 + 1], ids[n % 2:n % 2 + 1], False)
            tgt = torch.zeros(oh, ow, dtype=torch.bool)
            tgt[y0:y0 + oh // 3, x0:x0 + ow // 3] = True
            oI, oU, m, cam = O.eval_postprocess(o, tgt)
            assert (oI, oU) == (I, U)
            assert np.abs(cam[::8, ::8].numpy() - g[f"case{n}_cam_ds8"]).max() < TOL


def test_g10_pixel_attention_oracle_matches_reference(golden):
    """Stage-2 PixelAttention: oracle restatement vs the real reference module, forward + all gradients"""
    from oracle.gen_golden_data import pixel_attention_case
    g = golden("g10_pixel_attention.npz")
    N, Ci, Ct, H, W, T = (int(v) for v in g["dims"])
    sd, vis, lan = pixel_attention_case(3, N, Ci, Ct, H, W, T)
    sd = {"pa." + k: v.requires_grad_(True) for k, v in sd.items()}
    vis.requires_grad_(True)
    lan.requires_grad_(True)
    out = O.pixel_attention(sd, "pa", vis, lan)
    assert torch.allclose(out, torch.from_numpy(g["out"]), atol=1e-5, rtol=1e-5)
    out.backward(torch.from_numpy(g["gout"]))
    assert torch.allclose(vis.grad, torch.from_numpy(g["dvis"]), atol=1e-5, rtol=1e-4)
    assert torch.allclose(lan.grad, torch.from_numpy(g["dlan"]), atol=1e-5, rtol=1e-4)
    for k, v in sd.items():
        assert torch.allclose(v.grad, torch.from_numpy(g["d_" + k[3:]]), atol=1e-5, rtol=1e-4), k


def test_g11_vit_spatial_oracle_matches_reference_modules(golden):
    """dense ViT trunk (BASELINE config 5): oracle vs the reference's sub-modules glued by its commented variant"""
    from oracle.gen_golden_data import vit_case_state_dict
    g = golden("g11_vit_spatial.npz")
    sd = {"visual." + k: v.requires_grad_(True) for k, v in vit_case_state_dict().items()}
    cls, spa = O.encode_image_vit_spatial(sd, "", torch.from_numpy(g["img"]))
    assert torch.allclose(cls, torch.from_numpy(g["cls"]), atol=1e-4, rtol=1e-5)
    assert torch.allclose(spa, torch.from_numpy(g["spa"]), atol=1e-4, rtol=1e-5)
    ((spa * torch.from_numpy(g["gs"])).sum() + (cls * torch.from_numpy(g["gc"])).sum()).backward()
    for k in [n[2:] for n in g.files if n.startswith("d_") and not n.startswith("d_conv1")]:
        ref = torch.from_numpy(g["d_" + k])
        assert float((sd["visual." + k].grad - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6, k
    gw = sd["visual.conv1.weight"].grad
    assert abs(float(gw.norm()) - float(g["d_conv1.weight_norm"])) <= 1e-5 * float(g["d_conv1.weight_norm"])
