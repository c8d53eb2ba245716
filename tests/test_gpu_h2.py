"""The "h2" arithmetic (two fp16 pieces per operand -- the residual pre-scaled by 2^11 --, three f16 MFMAs per product, one
power-of-two scale per tensor from a device-side amax): the amax words, single products against fp64 next to the x3 / f32-MFMA
products, ADVERSARIAL operand statistics (tiny rows, outliers, per-channel gains, heavy tails), a heavy-tailed model against the
oracle, and training steps.  The whole parity suite (tests/test_gpu_parity.py) and every kernel test (tests/test_gpu_ops.py)
additionally run in h2 through their own parametrisation."""
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


def test_batchnorm_output_bound_word(ops):
    """tris_bn_out_bound_f32: max_c |gamma_c| sqrt(n - 1) + |beta_c| really bounds relu(bn(x)) (Samuelson), also for a
    heavy-tailed input; it is what scales the operand of a convolution whose BatchNorm input is never written"""
    import math
    g = torch.Generator().manual_seed(3)
    M, C = 4096, 64
    x = torch.distributions.StudentT(1.5).sample((M, C)).cuda().contiguous()       # (heavy tails: max |xhat| far above 5)
    gamma, beta = (torch.randn(C, generator=g) * 2).cuda(), torch.randn(C, generator=g).cuda()
    y = torch.relu(torch.nn.functional.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5))
    words = torch.zeros(WORD, device="cuda", dtype=torch.int32)
    ops.call("tris_bn_out_bound_f32", ops.P(gamma), ops.P(beta), C, math.sqrt(M - 1), words.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    bound = float(words.max().view(torch.float32))
    want = float((gamma.abs().double() * math.sqrt(M - 1) + beta.abs().double()).max())
    assert abs(bound - want) <= 1e-6 * want
    assert float(y.max()) <= bound
    assert float(y.max()) > 0.05 * bound       # (and for this input the bound is within a factor 20 of the true maximum)


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
    assert e_h2 < 2.0 * e_x3 + 1e-7, (e_h2, e_x3)
    assert e_after == e_x3
    if mag > 1.0 and K % 32 == 0 and M >= 4:      # unscaled, the same operands leave fp16's range: the scales matter
        ops.call("tris_h2_next", None, None, 1.0, 1.0)
        assert not (product() < 30.0 * e_x3)


# ---- adversarial operand statistics ("rows at 2^-20 amax, a single 1e4 outlier, log-uniform per-channel scales") ---
def _adversarial(kind, M, N, K, g):
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) * 0.1
    small = None      # rows of A / of B whose magnitude is far below the tensor's maximum: judged on their own
    if kind == "rows_2^-20":
        A[::2] *= 2.0 ** -20
        B[1::3] *= 2.0 ** -20
        small = (torch.arange(M) % 2 == 0, torch.arange(N) % 3 == 1)
    elif kind == "rows_2^-24":
        A[::2] *= 2.0 ** -24
        small = (torch.arange(M) % 2 == 0, None)
    elif kind == "outlier_1e4":
        A[5, 7] = 1e4
        B[3, 11] = -1e4
    elif kind == "outlier_1e7":
        A[5, 7] = 1e7
    elif kind == "channel_gains":                  # independent log-uniform per-channel (k) gains 2^-8 .. 2^8 on both operands
        A *= torch.exp2(torch.rand(K, generator=g) * 16 - 8)
        B *= torch.exp2(torch.rand(K, generator=g) * 16 - 8)
    elif kind == "channel_gains_anti":             # ... anti-correlated: every channel matters equally, across a 2^16 range
        ga = torch.exp2(torch.rand(K, generator=g) * 16 - 8)
        A *= ga
        B /= ga
    elif kind == "student_t":                      # heavy tails (2 degrees of freedom)
        A = torch.distributions.StudentT(2.0).sample((M, K))
        B = torch.distributions.StudentT(2.0).sample((N, K)) * 0.02
    elif kind == "gradient_like":                  # 1e-9-sized values with a few 1e-3 spikes
        A = torch.randn(M, K, generator=g) * 1e-9
        A[torch.randint(0, M, (20,), generator=g), torch.randint(0, K, (20,), generator=g)] = 1e-3
    return A.cuda().contiguous(), B.cuda().contiguous(), small


@pytest.mark.parametrize("kind", ["rows_2^-20", "rows_2^-24", "outlier_1e4", "outlier_1e7", "channel_gains", "channel_gains_anti",
                                  "student_t", "gradient_like"])
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
def test_h2_rowwise_error_on_adversarial_operands(kind, layout):
    """Row-wise (and column-wise) relative error of the h2 product against fp64, next to the f32-input MFMA's on the same
    operands: h2 must stay within 2x of it for EVERY statistic here -- including the rows that sit 2^-20 / 2^-24 below their
    tensor's maximum (the per-tensor scale would lose them with an unscaled residual piece: the reason lo' is stored x 2^11).
    Layouts: NT = Linear forward, NN = data gradient, TN = weight gradient (the reduction runs over A's rows)."""
    from tris_amd import ops
    M, N, K = 512, 384, 1024
    g = torch.Generator().manual_seed(sum(map(ord, kind)))
    A, B, small = _adversarial(kind, M, N, K, g)
    ref = A.double() @ B.double().t()
    if layout == "NT":
        a, b, tA, tB = A, B, False, True
    elif layout == "NN":
        a, b, tA, tB = A, B.t().contiguous(), False, False
    else:
        a, b, tA, tB = A.t().contiguous(), B.t().contiguous(), True, False
    C = torch.empty(M, N, device="cuda")
    prev = ops.get_gemm_mode()
    errs = {}
    try:
        for mode in ("f32", "x3", "h2"):
            ops.set_gemm_mode(mode)
            ops.gemm(a, b, C, M, N, K, a.shape[1], b.shape[1], N, tA, tB, use_ws=False)
            assert torch.isfinite(C).all(), mode
            d = C.double() - ref
            errs[mode] = (d.norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300), d.norm(dim=0) / ref.norm(dim=0).clamp_min(1e-300))
    finally:
        ops.set_gemm_mode(prev)
    for axis in (0, 1):
        eh, ef = errs["h2"][axis], errs["f32"][axis]
        sel = [torch.ones_like(eh, dtype=torch.bool)]
        if small is not None and small[axis] is not None:
            sel = [small[axis].cuda(), ~small[axis].cuda()]
        for s in sel:
            assert float(eh[s].max()) <= 2.0 * float(ef[s].max()), (kind, layout, axis, float(eh[s].max()), float(ef[s].max()))
            assert float(eh[s].median()) <= 2.0 * float(ef[s].median()), (kind, layout, axis)


def test_h2_error_floor_is_absolute_below_2_pow_minus_27():
    """the documented limit: rows 2^-32 below their tensor's maximum keep an ABSOLUTE accuracy of ~2^-49 of that maximum -- the
    result degrades gracefully (no garbage, no inf / NaN), it is no longer fp32-class for those rows"""
    from tris_amd import ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 256, 128, 512
    A = torch.randn(M, K, generator=g)
    A[::2] *= 2.0 ** -32
    A, B = A.cuda(), (torch.randn(N, K, generator=g) * 0.1).cuda()
    ref = A.double() @ B.double().t()
    C = torch.empty(M, N, device="cuda")
    prev = ops.get_gemm_mode()
    try:
        ops.set_gemm_mode("h2")
        ops.gemm(A, B, C, M, N, K, K, K, N, False, True)
    finally:
        ops.set_gemm_mode(prev)
    e = (C.double() - ref).norm(dim=1) / ref.norm(dim=1)
    assert float(e[1::2].max()) < 5e-7                       # the large rows: fp32-class
    assert 1e-7 < float(e[::2].max()) < 1e-3                 # the tiny rows: ~2^-49 / 2^-32 = 2^-17 relative, bounded


def _model_and_aux(fill, seed_m, seed_a):
    from tris_amd.args import get_parser
    from tris_amd.CLIP import clip
    from tris_amd.model.model_stage1 import TRIS
    from tris_amd.train_stage1 import freeze_aux
    args = get_parser().parse_args(["--size", "320", "--negative_samples", "3", "--max_query_len", "20"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = TRIS(args).cuda().train()
        aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
    fill(model.state_dict(), seed_m)
    fill(aux.state_dict(), seed_a)
    return args, model, freeze_aux(aux)


@pytest.mark.parametrize("arith", ["x3", "h2"])
def test_heavy_tailed_model_matches_the_oracle(arith):
    """Weights that look like released CLIP weights rather than iid noise -- log-uniform per-channel gains over 2^4, 1 % outlier
    channels x 16 on every weight matrix and norm scale (utils.synth.heavy_tail_fill) -- one Stage-1 step at B = 8 against the
    CPU oracle on the same state dict: losses, cls_out, sigmoid map within the north star's 1e-3, gradient arenas in direction
    and size.  Same tolerances in both arithmetics."""
    from oracle import tris_oracle as O
    from tris_amd import ops
    from tris_amd.optim import FusedAdamW
    from tris_amd.train_stage1 import stage1_forward_losses
    from tris_amd.utils.synth import heavy_tail_fill, synthetic_batch
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(arith)
    try:
        args, model, aux = _model_and_aux(heavy_tail_fill, 1234, 4321)
        B = 8
        b = synthetic_batch(B, 320, 20, 3, seed=11)
        sd = {k: v.detach().float().cpu().contiguous().clone() for k, v in model.state_dict().items()}
        auxsd = {k: v.detach().cpu().clone() for k, v in aux.state_dict().items()}
        bb, new = model.trainable_parameters()
        opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
        losses, cls, sig = stage1_forward_losses(model, aux, b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda(), args)
        opt.zero_grad()
        losses[0].backward()
        ops.wgrad_join()
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        leaves = O.trainable_split(sd)[0] + O.trainable_split(sd)[1]
        for k in leaves:
            sd[k].requires_grad_(True)
        ref = O.stage1_losses(sd, auxsd, b, faithful=False)
        ref["loss"].backward()
        got, want = losses.tolist(), [float(ref[k].detach()) for k in ("loss", "l1", "l4", "l5")]
        assert all(abs(a - c) < 1e-3 for a, c in zip(got, want)), (got, want)
        assert float((cls.detach().cpu() - ref["cls"].detach()).abs().max()) < 1e-3
        assert float((sig.detach().cpu() - ref["sig"].detach()).abs().max()) < 1e-3
        named = dict(model.named_parameters())
        dot = na = nb = 0.0
        for k in leaves:
            if sd[k].grad is None:
                continue
            a, c = named[k].grad.detach().double().cpu().reshape(-1), sd[k].grad.double().reshape(-1)
            dot, na, nb = dot + float(a @ c), na + float(a @ a), nb + float(c @ c)
        cos = dot / (na ** 0.5 * nb ** 0.5)
        assert cos > 0.998 and abs((na / nb) ** 0.5 - 1.0) < 1e-2, (cos, na, nb)
    finally:
        ops.set_gemm_mode(prev)


def _steps(arith, graph, B=3, steps=3):
    from tris_amd import ops
    from tris_amd.config import cfg
    from tris_amd.optim import FusedAdamW
    from tris_amd.train_stage1 import train_step
    from tris_amd.utils.synth import seed_fill, synthetic_batch
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(arith)
    try:
        with cfg.override(step_graph=graph):
            args, model, aux = _model_and_aux(seed_fill, 1234, 4321)
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
        ops.set_gemm_mode(prev)


def test_training_steps_in_h2_stay_on_the_fp32_noise_floor():
    """The whole step in h2.  The yardstick is the distance between the two fp32-class arithmetics the package already has (x3
    and the f32-input MFMA): h2 must sit within it, loss by loss and arena by arena; eager and segmented-graph replays of the h2
    step must agree bit for bit"""
    l3, g3, _ = _steps("x3", "0")
    lh, gh, used = _steps("h2", "0")
    assert used > 300                              # the step really tagged its operands
    lf, gf, _ = _steps("f32", "0")
    assert torch.isfinite(lh).all() and all(torch.isfinite(g).all() for g in gh)
    assert float((lh[0] - l3[0]).abs().max()) <= 3.0 * float((lf[0] - l3[0]).abs().max()) + 1e-4 * float(l3[0].abs().max())
    for a3, ah, af in zip(g3, gh, gf):
        assert float((ah - a3).norm()) <= 2.0 * float((af - a3).norm()) + 1e-6 * float(a3.norm())
    lg, gg, _ = _steps("h2", "seg")
    assert torch.equal(lg, lh)
    for a, b in zip(gg, gh):
        assert torch.equal(a, b)
