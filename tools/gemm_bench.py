"""Dev tool: micro-benchmark of the GEMM / conv core on representative shapes."""
import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
torch.manual_seed(0)
def bench(fn, flops, name, it=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print(f"{name:46s} {ms*1e3:9.1f} us  {flops/(ms*1e-3)/1e12:7.2f} TF/s", flush=True)
def g(M, N, K, tA=False, tB=True):
    A = torch.randn((K, M) if tA else (M, K), device="cuda"); B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    bench(lambda: ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, tA, tB), 2.0*M*N*K, f"gemm {'T' if tA else 'N'}{'T' if tB else 'N'} M{M} N{N} K{K}")
g(4096, 4096, 4096); g(8192, 8192, 1024); g(2400, 768, 3072); g(2400, 3072, 768); g(3840, 512, 2048); g(307200, 256, 64); g(307200, 64, 256, False, False); g(76800, 512, 128); g(19200, 1024, 256); g(4800, 2048, 512)
g(256, 64, 307200, True, False); g(1024, 256, 19200, True, False); g(512, 512, 960, True, False); g(960, 512, 512)
def conv(B, H, C1, C2):
    x = torch.randn(B, H, H, C1, device="cuda"); w = torch.randn(C2, C1, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, H, H, C2, device="cuda"); dw = torch.empty_like(w); ws = ops.workspace(0)
    fl = 2.0*B*H*H*C2*9*C1
    bench(lambda: ops.call("tris_conv3x3_fwd_f32", ops.P(x), ops.P(w), ops.P(y), B, H, H, C1, C2, 1, ops._stream()), fl, f"conv fwd B{B} {H}x{H} {C1}->{C2}")
    bench(lambda: ops.call("tris_conv3x3_dgrad_f32", ops.P(y), ops.P(w), ops.P(x), B, H, H, C1, C2, ops._stream()), fl, f"conv dgrad B{B} {H}x{H} {C1}->{C2}")
    bench(lambda: ops.call("tris_conv3x3_wgrad_f32", ops.P(x), ops.P(y), ops.P(dw), B, H, H, C1, C2, 1, ops.P(ws), ws.numel()*4, ops._stream()), fl, f"conv wgrad B{B} {H}x{H} {C1}->{C2}")
conv(48, 80, 64, 64); conv(48, 80, 128, 128); conv(48, 40, 256, 256); conv(48, 20, 512, 512); conv(48, 10, 512, 512); conv(48, 160, 32, 64)
