"""h2 products on fp32 operands (split in the kernel, PREC 3) against the same products on operand planes (PREC 4: the pieces arrive
split, csrc/planes.h), per shape on an idle device: time of both, and that the results are bit-identical.
usage: python tools/planes_bench.py [quick]      (autotuned: every form gets its own best tile / split-K / loop)"""
import sys
import torch
sys.path.insert(0, ".")
from tris_amd import ops
from tris_amd.ops import P, call, _stream

torch.manual_seed(0)
DEV = "cuda"


def word_of(t):
    w = torch.zeros(2048, dtype=torch.int32, device=DEV)
    call("tris_amax_bits_f32", P(t), t.numel(), w.data_ptr(), _stream())
    return w


def planes_of(t, w):
    o = torch.empty_like(t)
    call("tris_h2_planes_f32", P(t), P(o), t.numel(), w.data_ptr(), _stream())
    return o


def timed(fn, it=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3   # us


def report(name, flops, run3, run4, out3, out4):
    t3 = timed(run3)
    r3 = out3().clone()
    t4 = timed(run4)
    r4 = out4().clone()
    same = torch.equal(r3, r4)
    md = (r3 - r4).abs().max().item()
    print(f"{name:44s} split-in-kernel {t3:8.1f} us {flops / t3 * 1e-6:6.1f} TF/s | planes {t4:8.1f} us {flops / t4 * 1e-6:6.1f} TF/s "
          f"| x{t3 / t4:4.2f} | {'bit-identical' if same else f'DIFFERENT max {md:.3e}'}", flush=True)


def gemm(M, N, K, tA=False, tB=True):
    A = torch.randn((K, M) if tA else (M, K), device=DEV)
    B = torch.randn((N, K) if tB else (K, N), device=DEV)
    wa, wb = word_of(A), word_of(B)
    Ap, Bp = planes_of(A, wa), planes_of(B, wb)
    C3, C4 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ws = ops.workspace(0)

    def go(a, b, c, planes):
        if planes:
            call("tris_h2_next_planes", wa.data_ptr(), wb.data_ptr(), 0)
        else:
            call("tris_h2_next", wa.data_ptr(), wb.data_ptr(), 0.0, 0.0)
        call("tris_gemm_f32", P(a), P(b), P(c), M, N, K, a.shape[1], b.shape[1], N, int(tA), int(tB), 1, 0, 0, 0, None, 0, None, 0, 0,
             0, 1.0, P(ws), ws.numel() * 4, _stream())
    report(f"gemm {'T' if tA else 'N'}{'T' if tB else 'N'} M{M} N{N} K{K}", 2.0 * M * N * K, lambda: go(A, B, C3, False),
           lambda: go(Ap, Bp, C4, True), lambda: C3, lambda: C4)


def conv(B, H, C1, C2):
    x = torch.randn(B, H, H, C1, device=DEV)
    w = torch.randn(C2, 3, 3, C1, device=DEV)      # [Cout][kh][kw][Cin] as it lies in memory
    dy = torch.randn(B, H, H, C2, device=DEV)
    wx, ww, wd = word_of(x), word_of(w), word_of(dy)
    xp, wp, dp = planes_of(x, wx), planes_of(w, ww), planes_of(dy, wd)
    fl = 2.0 * B * H * H * C2 * 9 * C1
    ws = ops.workspace(0)
    y3, y4 = torch.empty(B, H, H, C2, device=DEV), torch.empty(B, H, H, C2, device=DEV)

    def arm(a, b, planes):
        if planes:
            call("tris_h2_next_planes", a.data_ptr(), b.data_ptr(), 0)
        else:
            call("tris_h2_next", a.data_ptr(), b.data_ptr(), 0.0, 0.0)

    def fwd(xx, wwt, y, planes):
        arm(wx, ww, planes)
        call("tris_conv3x3_fwd_f32", P(xx), P(wwt), P(y), B, H, H, C1, C2, 1, _stream())
    report(f"conv3x3 fwd   B{B} {H}x{H} {C1}->{C2}", fl, lambda: fwd(x, w, y3, False), lambda: fwd(xp, wp, y4, True), lambda: y3, lambda: y4)
    dx3, dx4 = torch.empty_like(x), torch.empty_like(x)

    def dgrad(d, wwt, o, planes):
        arm(wd, ww, planes)
        call("tris_conv3x3_dgrad_f32", P(d), P(wwt), P(o), B, H, H, C1, C2, _stream())
    report(f"conv3x3 dgrad B{B} {H}x{H} {C1}->{C2}", fl, lambda: dgrad(dy, w, dx3, False), lambda: dgrad(dp, wp, dx4, True), lambda: dx3,
           lambda: dx4)
    dw3, dw4 = torch.empty_like(w), torch.empty_like(w)

    def wgrad(xx, d, o, planes):
        arm(wd, wx, planes)
        call("tris_conv3x3_wgrad_f32", P(xx), P(d), P(o), B, H, H, C1, C2, 1, P(ws), ws.numel() * 4, _stream())
    report(f"conv3x3 wgrad B{B} {H}x{H} {C1}->{C2}", fl, lambda: wgrad(x, dy, dw3, False), lambda: wgrad(xp, dp, dw4, True), lambda: dw3,
           lambda: dw4)


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
if len(sys.argv) > 1 and sys.argv[1] == "vit":
    # round 6: the frozen aux ViT-B/32's products at the headline batch (48 x 50 tokens): would "convert A to planes, then the plane
    # product on once-converted weight planes" beat the product that splits both operands in its loop?  (+ the cost of the conversion)
    for n in (768, 2304, 3072):
        x = torch.randn(2400, n, device=DEV)
        w = word_of(x)
        o = torch.empty_like(x)
        print(f"convert [2400, {n}] fp32 -> planes: {timed(lambda: call('tris_h2_planes_f32', P(x), P(o), x.numel(), w.data_ptr(), _stream())):6.1f} us"
              f"   amax pre-pass: {timed(lambda: call('tris_amax_bits_f32', P(x), x.numel(), w.data_ptr(), _stream())):6.1f} us", flush=True)
    for (M, N, K) in [(2400, 2304, 768), (2400, 768, 768), (2400, 3072, 768), (2400, 768, 3072), (2400, 768, 2304), (3840, 1536, 512),
                      (3840, 512, 512), (3840, 2048, 512), (3840, 512, 2048)]:
        gemm(M, N, K)
    for (M, N, K) in [(2400, 768, 2304), (2400, 768, 768), (2400, 768, 3072), (2400, 3072, 768)]:
        gemm(M, N, K, False, False)
    sys.exit(0)
# the four shapes VERDICT r4 names, then the trunk's 1x1 products (forward NT, data gradient NN, weight gradient TN) and the 3x3 stages
gemm(76800, 256, 2304)
gemm(19200, 512, 4608)
gemm(19200, 1024, 256)
gemm(4096, 4096, 4096)
if not quick:
    for (M, N, K) in [(307200, 64, 256), (307200, 256, 64), (76800, 512, 128), (76800, 128, 512), (76800, 512, 256), (19200, 1024, 512),
                      (19200, 256, 1024), (4800, 2048, 1024), (4800, 512, 2048), (4800, 2048, 512), (4800, 1024, 2048)]:
        gemm(M, N, K)
    for (M, N, K) in [(307200, 64, 256), (76800, 256, 512), (19200, 512, 1024), (19200, 256, 1024), (4800, 1024, 2048), (4800, 512, 2048)]:
        gemm(M, N, K, False, False)
    for (M, N, K) in [(256, 64, 307200), (512, 128, 76800), (1024, 256, 19200), (512, 1024, 19200), (2048, 512, 4800), (1024, 2048, 4800)]:
        gemm(M, N, K, True, False)
conv(48, 80, 128, 128)
conv(48, 40, 256, 256)
if not quick:
    conv(48, 80, 64, 64)
    conv(48, 40, 128, 128)
    conv(48, 20, 256, 256)
    conv(48, 20, 512, 512)
    conv(48, 10, 512, 512)
    conv(48, 160, 32, 32)
    conv(48, 160, 32, 64)
