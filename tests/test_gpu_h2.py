"""The opt-in "h2" arithmetic (two fp16 pieces per operand, three f16 MFMAs per product, power-of-two operand scales from a
device-side amax): accuracy of single products against fp64 next to the x3 default, the amax words, and a training step."""
import os
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

WORD = 2048   # unsigned words per amax word (include/tris_hip.h)


def _rel(c, ref):
    return float((c.double() - ref).norm() / ref.norm())


@pytest.fixture()
def ops():
    from tris_amd import ops as o
    prev = o.get_gemm_mode()
    o.set_gemm_mode("x3")
    yield o
    o.set_gemm_mode(prev)


@pytest.mark.parametrize("n", [1, 7, 4096, (1 << 20) + 3, 5 << 20])
def test_amax_words_hold_the_largest_magnitude(ops, n):
    torch.manual_seed(n)
    x = torch.randn(n + 4, device="cuda")[:n] if n % 4 else torch.randn(n, device="cuda")
    x = x.contiguous()
    words = torch.zeros(WORD, device="cuda", dtype=torch.int32)
    ops.call("tris_amax_bits_f32", ops.P(x), n, words.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    got = float(words.max().view(torch.float32))
    assert got == float(x.abs().max())
    used = torch.nonzero(words).flatten().tolist()
    assert all(i % 16 == 0 for i in used)        # one word per cache line


@pytest.mark.parametrize("M,N,K,tA,tB", [(4096, 1024, 512, False, True), (2400, 768, 3072, False, True), (130, 68, 96, False, True),
                                         (1024, 256, 4800, True, False), (960, 512, 512, False, False)])
@pytest.mark.parametrize("mag", [1.0, 1e-6, 3e4])
def test_scaled_h2_product_is_as_accurate_as_x3(ops, M, N, K, tA, tB, mag):
    """operand magnitudes far outside fp16's range: with the device-side scales the h2 product has the error of x3 / fp32;
    armed for ONE product only"""
    torch.manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), device="cuda") * mag
    B = torch.randn((N, K) if tB else (K, N), device="cuda") * 0.1
    ref = (A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double())
    C = torch.empty(M, N, device="cuda")
    lda, ldb = A.shape[1], B.shape[1]

    def product():
        ops.gemm(A, B, C, M, N, K, lda, ldb, N, tA, tB)
        return _rel(C, ref)
    e_x3 = product()
    words = torch.zeros(2 * WORD, device="cuda", dtype=torch.int32)
    ops.call("tris_amax_bits_f32", ops.P(A), A.numel(), words.data_ptr(), ops._stream())
    ops.call("tris_amax_bits_f32", ops.P(B), B.numel(), words.data_ptr() + 4 * WORD, ops._stream())
    ops.call("tris_h2_next", words.data_ptr(), words.data_ptr() + 4 * WORD, 0.0, 0.0)
    e_h2 = product()
    e_after = product()                           # not armed any more
    assert torch.isfinite(C).all()
    assert e_h2 < 3.0 * e_x3 + 1e-7, (e_h2, e_x3)
    assert e_after == e_x3
    if mag != 1.0 and K % 32 == 0 and M >= 4:     # unscaled, the same operands leave fp16's range: the scales matter
        ops.call("tris_h2_next", None, None, 1.0, 1.0)
        assert not (product() < 30.0 * e_x3)


def _steps(linear_mode, graph, B=3, steps=3):
    from tris_amd import ops
    from tris_amd.args import get_parser
    from tris_amd.CLIP import clip
    from tris_amd.model.model_stage1 import TRIS
    from tris_amd.optim import FusedAdamW
    from tris_amd.train_stage1 import freeze_aux, train_step
    from tris_amd.utils.synth import seed_fill, synthetic_batch
    old = {k: os.environ.get(k) for k in ("TRIS_LINEAR_MODE", "TRIS_STEP_GRAPH")}
    os.environ["TRIS_LINEAR_MODE"], os.environ["TRIS_STEP_GRAPH"] = linear_mode, graph
    try:
        args = get_parser().parse_args(["--size", "320", "--negative_samples", "3", "--max_query_len", "20"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = TRIS(args).cuda().train()
            aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
        seed_fill(model.state_dict(), 1234)
        seed_fill(aux.state_dict(), 4321)
        freeze_aux(aux)
        bb, new = model.trainable_parameters()
        opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
        losses, g1 = [], None
        for s in range(steps):
            b = synthetic_batch(B, 320, 20, 3, seed=7 + s)
            out = train_step(model, aux, opt, b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda(), args, None)
            losses.append(out.clone())
            if s == 0:
                torch.cuda.synchronize()
                g1 = [a.g.clone() for a in opt.arenas]
        torch.cuda.synchronize()
        return torch.stack(losses), g1, ops._H2["next"]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_training_steps_with_h2_linear_products_stay_on_the_fp32_noise_floor():
    """TRIS_LINEAR_MODE=h2: Linear / 1x1 products of the step in h2.  The yardstick is the distance between the two fp32-class
    arithmetics the package already has (x3 and the f32-input MFMA): h2 must sit within it, loss by loss and arena by arena;
    eager and segmented-graph replays of the h2 step must agree bit for bit"""
    from tris_amd import ops
    l3, g3, used3 = _steps("", "0")
    assert used3 == 0 or not ops._H2["live"]
    lh, gh, used = _steps("h2", "0")
    assert used > 300                              # the step really tagged its operands
    ops.set_gemm_mode("f32")
    try:
        lf, gf, _ = _steps("", "0")
    finally:
        ops.set_gemm_mode("x3")
    assert torch.isfinite(lh).all() and all(torch.isfinite(g).all() for g in gh)
    assert float((lh[0] - l3[0]).abs().max()) <= 3.0 * float((lf[0] - l3[0]).abs().max()) + 1e-4 * float(l3[0].abs().max())
    for a3, ah, af in zip(g3, gh, gf):
        assert float((ah - a3).norm()) <= 2.0 * float((af - a3).norm()) + 1e-6 * float(a3.norm())
    lg, gg, _ = _steps("h2", "seg")
    assert torch.equal(lg, lh)
    for a, b in zip(gg, gh):
        assert torch.equal(a, b)
