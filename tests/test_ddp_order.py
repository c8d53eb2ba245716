"""Host-only test of the data-parallel reducer's ORDER invariant on the real Stage-1 model code.

The real `TRIS.forward` / `ModifiedResNet.forward_cl` / `CLIP.encode_text` run on CPU with the kernels of
`tris_amd.ops` replaced by small differentiable stand-ins (numerics are irrelevant here): what is under test is the
autograd graph the product code builds -- its issue order, the places where it calls `GradReducer.boundary`, the
segment plan (`STAGE1_RULES`) -- and the rule "no segment is all-reduced before the gradient of every parameter in
it has been written".  Round 1 broke that rule for the whole text encoder: forward issues the
text encoder BEFORE the trunk when it runs on the side stream, so its backward runs AFTER the stem's, yet its
parameters rode on the boundary behind layer4.  The same invariant is checked on the GPU with NaN-poisoned arenas
(GradReducer(check=True), tests/test_gpu_parity.py, tests/test_gpu_ddp.py).
"""
import contextlib
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F


def _install_standins(mp):
    from tris_amd import ops

    def nchw_to_nhwc(x):
        return x.permute(0, 2, 3, 1).contiguous()

    def conv3x3(x, w, stride=1, stats=False):
        return F.conv2d(x.permute(0, 3, 1, 2), w, stride=stride, padding=1).permute(0, 2, 3, 1).contiguous()

    def linear(x, w, b=None, resid=None, act=0, stats=False, grad_box=None, grad_box_out=None, grad_box_res=None, act_link=False):
        y = x @ w.reshape(w.shape[0], -1).t()
        if b is not None:
            y = y + b
        if act:
            y = torch.relu(y)
        return y if resid is None else y + resid

    def batch_norm(x, g, b, rm, rv, resid=None, relu=False, training=True, momentum=0.1, eps=1e-5, group=None,
                   grad_box=None, lazy=False, bwd_link=False, pool=False, planes=False, dx_planes=False):
        y = x * g + b          # (statistics are irrelevant for the order of the graph)
        if resid is not None:
            y = y + resid
        y = torch.relu(y) if relu else y
        return F.avg_pool2d(y.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous() if pool else y

    def avgpool2(x, grad_box_out=None):
        return F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous()

    def embed(ids, tok, pos, reducer=None):   # (dense stand-in: the sparse row exchange is exercised on the GPU, test_gpu_ddp)
        return tok[ids] + pos[:ids.shape[1]]

    def layer_norm(x, g, b, eps=1e-5, grad_box=None):
        return F.layer_norm(x, (x.shape[-1],), g, b, eps)

    def mha(qkv, heads, causal):
        W = qkv.shape[-1] // 3
        return qkv[..., :W] + qkv[..., W:2 * W] + qkv[..., 2 * W:]

    def eot_gather(ids, x):
        return x[torch.arange(x.shape[0]), ids.argmax(-1)]

    def matmul(A, B, tB=False):
        return A @ (B.t() if tB else B)

    def bmm(A, B, tB=False, alpha=1.0):
        return alpha * (A @ (B.transpose(1, 2) if tB else B))

    def l2norm(x):
        return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-6)

    def instance_norm(x, g, b, relu=False, eps=1e-5):
        y = x * g + b
        return torch.relu(y) if relu else y

    def xattn(Qv, Kv, Vv, Qt, Kt, Vt):
        t = (Qt + Kt + Vt).mean(0)
        v = (Qv + Kv + Vv)
        return v + t, v.mean(1, keepdim=True) + (Qt + Kt + Vt)[None]

    def score_heads(score, h, w, S, train, focal_p=3.0, focal_c=0.01):
        B = score.shape[0]
        m = score.mean((1, 2)).view(B, 1, 1, 1).expand(B, 1, S, S)
        if not train:
            return torch.relu(m)
        cls = score.mean(1)
        return cls, cls.diagonal().detach(), torch.relu(m), torch.sigmoid(m)

    for name, fn in dict(nchw_to_nhwc=nchw_to_nhwc, conv3x3=conv3x3, conv3x3_bnin_ok=lambda shape, cout: False, linear=linear,
                         batch_norm=batch_norm,
                         avgpool2=avgpool2, embed=embed, layer_norm=layer_norm, mha=mha, eot_gather=eot_gather,
                         matmul=matmul, bmm=bmm, l2norm=l2norm, instance_norm=instance_norm, xattn=xattn,
                         score_heads=score_heads, quick_gelu=torch.sigmoid, axpy=lambda a, b, s, grad_box_b=None: s * a + b,
                         axpy_bcast=lambda a, b, s, grad_box_b=None: s * a + b, scale_exp=lambda x, ls: (x * ls.exp(), ls.exp()),
                         linear_qgelu=lambda x, w, b=None: torch.sigmoid(linear(x, w, b))).items():
        mp.setattr(ops, name, fn)


class _FakeStream:
    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass


def _emulated_arenas(m):
    bb, new = m.trainable_parameters()
    arenas = []
    for group in (bb, new):
        ps = [p for p in group if not getattr(p, "_tris_no_grad_path", False)]
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 63) // 64 * 64
        arenas.append(SimpleNamespace(params=ps, offsets=offs, numel=n, g=torch.zeros(1)))
    return arenas


class _FakeEvent:
    def record(self, stream=None):
        pass


@pytest.mark.parametrize("overlap", [True, "mid", False], ids=["text-issued-first", "text-issued-mid-trunk", "serial"])
def test_no_segment_is_reduced_before_its_last_gradient(monkeypatch, overlap):
    from tris_amd import comm, parallel
    from tris_amd.model import model_stage1
    from tris_amd.utils.shapes import _build_tris
    _install_standins(monkeypatch)
    monkeypatch.setattr(model_stage1, "_overlap_enabled", lambda: bool(overlap))
    monkeypatch.setattr(model_stage1.cfg, "text_at", "layer2" if overlap == "mid" else "start")
    monkeypatch.setattr(model_stage1, "_side_stream", lambda dev: _FakeStream())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _FakeEvent())
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None, raising=False)
    reduced = []
    monkeypatch.setattr(comm, "all_reduce", lambda t, op=None, group=None, async_op=False: (reduced.append(t.numel()), comm._Done())[1])

    m = _build_tris().train()
    arenas = _emulated_arenas(m)
    red = parallel.GradReducer([torch.zeros(a.numel) for a in arenas], force=True, check=False)
    seg, par = parallel.stage1_segments(m, SimpleNamespace(arenas=arenas), with_params=True)
    red.set_segments(seg, par)
    m.backbone.visual.grad_reducer = red
    m.backbone.grad_reducer = red

    # every arena slot is covered exactly once, every parameter sits in exactly one segment
    cover = [0, 0]
    for ranges in seg.values():
        for ai, s, e in ranges:
            cover[ai] += e - s
    assert cover == [a.numel for a in arenas]
    names = [n for ranges in par.values() for *_, n in ranges]
    trainable = {n for n, p in m.named_parameters()
                 if not getattr(p, "_tris_no_grad_path", False) and n != "logit_scale"}
    assert sorted(names) == sorted(trainable)
    # the text encoder is NOT in the segment released behind layer4 (the round-1 defect)
    assert all(not n.startswith("backbone.") for *_, n in par["heads"])
    assert any(n.startswith("backbone.transformer.resblocks.0.") for *_, n in par["text"])
    assert any(n.startswith("backbone.transformer.resblocks.11.") for *_, n in par["text_hi"])
    assert any(n.startswith("backbone.transformer.resblocks.5.") for *_, n in par["text_mid"])
    assert "backbone.text_projection" in {n for *_, n in par["text_hi"]}
    assert {n for *_, n in par["embed"]} == {"backbone.token_embedding.weight", "backbone.positional_embedding"}

    written = set()
    name_of = {id(p): n for n, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
        p.register_post_accumulate_grad_hook(lambda q: written.add(name_of[id(q)]))
    launches = []
    orig = red._launch_now      # (boundary nodes call _launch, which defers to _launch_now outside a graph capture; finish() calls it directly)

    def checked(key):
        if key not in red.done:
            missing = [n for *_, n in red.param_ranges[key] if n not in written]
            assert not missing, f"segment {key!r} released before {missing[:3]} (+{len(missing)}) had a gradient"
            launches.append(key)
        orig(key)
    red._launch_now = checked

    B = 2
    img = torch.randn(B, 3, 64, 64)
    ids = torch.randint(1, 1000, (B, 20))
    cls, fg, relu_map, sig, ls = m(img, ids)
    red.begin_step()
    (cls.sum() + sig.mean() + relu_map.mean()).backward()
    in_backward = list(launches)
    red.finish()
    assert sorted(launches) == sorted(seg)                       # every segment reduced, once
    assert sum(reduced) == sum(a.numel for a in arenas)          # ... and every arena element with it
    # overlap with backward: the trunk stages and the heads are released from inside backward, in completion order
    order = [k for k in in_backward if k in ("heads", "layer4", "layer3", "layer2", "layer1")]
    assert order == ["heads", "layer4", "layer3", "layer2", "layer1"]
    assert "text" in in_backward and "embed" not in in_backward and "stem" not in in_backward
    assert [k for k in in_backward if k.startswith("text")] == ["text_hi", "text_mid", "text"]
    if overlap == "mid":   # issued behind layer2 => its backward sits between layer3's and layer2's
        assert in_backward.index("layer4") < in_backward.index("text_hi") and in_backward.index("text") < in_backward.index("layer2")
    elif overlap:          # issued first => its backward (and its release) come after the whole trunk
        assert in_backward.index("text_hi") > in_backward.index("layer1")
    else:                  # issued after the trunk => released before the trunk starts
        assert in_backward.index("text") < in_backward.index("layer4")
    assert len(written) >= len(trainable)


def test_round1_plan_would_have_been_caught(monkeypatch):
    """the round-1 rule set (text encoder in the segment released behind layer4) trips the check"""
    from tris_amd import comm, parallel
    from tris_amd.model import model_stage1
    from tris_amd.utils.shapes import _build_tris
    _install_standins(monkeypatch)
    monkeypatch.setattr(model_stage1, "_overlap_enabled", lambda: True)
    monkeypatch.setattr(model_stage1.cfg, "text_at", "start")
    monkeypatch.setattr(model_stage1, "_side_stream", lambda dev: _FakeStream())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _FakeEvent())
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None, raising=False)
    monkeypatch.setattr(comm, "all_reduce", lambda t, op=None, group=None, async_op=False: comm._Done())
    m = _build_tris().train()
    arenas = _emulated_arenas(m)
    old_rules = {"heads": lambda n: not n.startswith("backbone.visual."),
                 "layer4": lambda n: n.startswith("backbone.visual.layer4."), "stem": lambda n: True}
    red = parallel.GradReducer([torch.zeros(a.numel) for a in arenas], force=True, check=False)
    red.set_segments(*parallel.GradReducer.plan(arenas, list(m.named_parameters()), old_rules, with_params=True))
    m.backbone.visual.grad_reducer = red
    written = set()
    name_of = {id(p): n for n, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
        p.register_post_accumulate_grad_hook(lambda q: written.add(name_of[id(q)]))
    bad = {}
    orig = red._launch_now      # (boundary nodes call _launch, which defers to _launch_now outside a graph capture; finish() calls it directly)

    def spy(key):
        if key not in red.done:
            bad[key] = [n for *_, n in red.param_ranges.get(key, []) if n not in written]
        orig(key)
    red._launch = spy
    cls, fg, relu_map, sig, ls = m(torch.randn(2, 3, 64, 64), torch.randint(1, 1000, (2, 20)))
    (cls.sum() + sig.mean()).backward()
    assert any(n.startswith("backbone.transformer.") for n in bad["heads"])
