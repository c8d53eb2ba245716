"""Dev tool: direct 3x3 convolution (A_HALO kernels) against the implicit GEMM on the trunk's shapes (idle device)."""
import os, sys, torch
sys.path.insert(0, ".")
os.environ.setdefault("TRIS_RANDOM_INIT", "1")
from tris_amd import ops
torch.manual_seed(0)
def bench(fn, it=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
def conv(B, H, C1, C2):
    x = torch.randn(B, H, H, C1, device="cuda"); w = torch.randn(C2, C1, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last) * 0.05
    y = torch.empty(B, H, H, C2, device="cuda"); dx = torch.empty_like(x)
    fl = 2.0*B*H*H*C2*9*C1
    f = lambda: ops.call("tris_conv3x3_fwd_f32", ops.P(x), ops.P(w), ops.P(y), B, H, H, C1, C2, 1, ops._stream())
    d = lambda: ops.call("tris_conv3x3_dgrad_f32", ops.P(y), ops.P(w), ops.P(dx), B, H, H, C1, C2, ops._stream())
    for name, fn, ref in (("fwd", f, y), ("dgrad", d, dx)):
        row = f"{name:5s} B{B} {H}x{H} {C1:4d}->{C2:4d} "
        os.environ["TRIS_CONV_DIRECT"] = "0"
        fn(); base = ref.clone()
        ms = bench(fn)
        row += f"| implicit {ms*1e3:7.1f} us {fl/(ms*1e-3)/1e12:6.1f} TF "
        for i in range(1, 7):
            os.environ["TRIS_CONV_DIRECT"] = str(i)
            ref.zero_(); fn(); torch.cuda.synchronize()
            err = float((ref - base).abs().max()) / max(1e-30, float(base.abs().max()))
            if err > 1e-5:
                row += f"| d{i} WRONG {err:.1e} "
                continue
            ms2 = bench(fn)
            if abs(ms2 - ms) / ms < 0.02 and i > 0:
                pass
            row += f"| d{i} {ms2*1e3:7.1f} us {fl/(ms2*1e-3)/1e12:6.1f} TF "
        print(row, flush=True)
    os.environ.pop("TRIS_CONV_DIRECT", None)
for s in [(48, 160, 32, 32), (48, 160, 32, 64), (48, 80, 64, 64), (48, 80, 128, 128), (48, 40, 128, 128), (48, 40, 256, 256),
          (48, 20, 256, 256), (48, 20, 512, 512), (48, 10, 512, 512)]:
    conv(*s)
print("---- weight gradient", flush=True)
def wg(B, H, C1, C2):
    x = torch.randn(B, H, H, C1, device="cuda"); dy = torch.randn(B, H, H, C2, device="cuda")
    dw = torch.empty(C2, 3, 3, C1, device="cuda"); ws = ops.workspace(0)
    fl = 2.0*B*H*H*C2*9*C1
    fn = lambda: ops.call("tris_conv3x3_wgrad_f32", ops.P(x), ops.P(dy), ops.P(dw), B, H, H, C1, C2, 1, ops.P(ws), ws.numel()*4, ops._stream())
    row = f"wgrad B{B} {H}x{H} {C1:4d}->{C2:4d} "
    os.environ["TRIS_WGRAD_DIRECT"] = "0"
    fn(); base = dw.clone(); ms = bench(fn)
    row += f"| implicit {ms*1e3:7.1f} us {fl/(ms*1e-3)/1e12:6.1f} TF "
    for i in range(1, 6):
        os.environ["TRIS_WGRAD_DIRECT"] = str(i)
        dw.zero_(); fn(); torch.cuda.synchronize()
        err = float((dw - base).abs().max()) / max(1e-30, float(base.abs().max()))
        if err > 1e-5:
            row += f"| w{i} WRONG {err:.1e} "; continue
        ms2 = bench(fn)
        row += f"| w{i} {ms2*1e3:7.1f} us {fl/(ms2*1e-3)/1e12:6.1f} TF "
    print(row, flush=True)
    os.environ.pop("TRIS_WGRAD_DIRECT", None)
for s in [(48, 160, 32, 32), (48, 160, 32, 64), (48, 80, 64, 64), (48, 80, 128, 128), (48, 40, 128, 128), (48, 40, 256, 256)]:
    wg(*s)
