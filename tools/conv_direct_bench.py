"""Dev tool: the 3x3 convolutions of the trunk's shapes on an idle device -- every direct configuration (A_HALO kernels, direct
weight gradient) against the implicit GEMM, in both arithmetics (x3, h2).  usage: python tools/conv_direct_bench.py [x3|h2 ...]"""
import os, sys, torch
sys.path.insert(0, ".")
os.environ.setdefault("TRIS_RANDOM_INIT", "1")
from tris_amd import ops
torch.manual_seed(0)
MODES = sys.argv[1:] or ["x3", "h2"]
def bench(fn, it=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
def sweep(name, fn, out, fl, opt, n):
    for mode in MODES:
        ops.set_gemm_mode(mode)
        row = f"{name} [{mode}] "
        ops.set_option(opt, 0)
        fn(); base = out.clone(); ms = bench(fn)
        row += f"| implicit {ms*1e3:7.1f} us {fl/(ms*1e-3)/1e12:6.1f} TF "
        for i in range(1, n):
            ops.set_option(opt, i)
            out.zero_(); fn(); torch.cuda.synchronize()
            err = float((out - base).abs().max()) / max(1e-30, float(base.abs().max()))
            if err > 1e-5:
                row += f"| {i}: n/a " if err > 0.5 else f"| {i}: WRONG {err:.1e} "
                continue
            ms2 = bench(fn)
            if abs(ms2 - ms) / ms > 0.02:
                row += f"| {i}: {ms2*1e3:7.1f} us {fl/(ms2*1e-3)/1e12:6.1f} TF "
        print(row, flush=True)
    ops.set_option(opt, None)
def conv(B, H, C1, C2):
    x = torch.randn(B, H, H, C1, device="cuda"); w = (torch.randn(C2, C1, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, H, H, C2, device="cuda"); dx = torch.empty_like(x)
    fl = 2.0*B*H*H*C2*9*C1
    f = lambda: (ops.h2_arm(x, w), ops.call("tris_conv3x3_fwd_f32", ops.P(x), ops.P(w), ops.P(y), B, H, H, C1, C2, 1, ops._stream()))
    d = lambda: (ops.h2_arm(y, w), ops.call("tris_conv3x3_dgrad_f32", ops.P(y), ops.P(w), ops.P(dx), B, H, H, C1, C2, ops._stream()))
    f()
    sweep(f"fwd   B{B} {H}x{H} {C1:4d}->{C2:4d}", f, y, fl, "CONV_DIRECT", 7)
    sweep(f"dgrad B{B} {H}x{H} {C1:4d}->{C2:4d}", d, dx, fl, "CONV_DIRECT", 7)
def wg(B, H, C1, C2):
    x = torch.randn(B, H, H, C1, device="cuda"); dy = torch.randn(B, H, H, C2, device="cuda")
    dw = torch.empty(C2, 3, 3, C1, device="cuda"); ws = ops.workspace(0)
    fl = 2.0*B*H*H*C2*9*C1
    fn = lambda: (ops.h2_arm(dy, x), ops.call("tris_conv3x3_wgrad_f32", ops.P(x), ops.P(dy), ops.P(dw), B, H, H, C1, C2, 1, ops.P(ws), ws.numel()*4, ops._stream()))
    sweep(f"wgrad B{B} {H}x{H} {C1:4d}->{C2:4d}", fn, dw, fl, "WGRAD_DIRECT", 6)
SH = [(48, 160, 32, 32), (48, 160, 32, 64), (48, 80, 64, 64), (48, 80, 128, 128), (48, 40, 128, 128), (48, 40, 256, 256),
      (48, 20, 256, 256), (48, 20, 512, 512), (48, 10, 512, 512)]
for s in SH: conv(*s)
print("---- weight gradient", flush=True)
for s in SH: wg(*s)
