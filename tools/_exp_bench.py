import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
def bench(fn, flops, name, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print(f"{name:40s} {ms*1e3:9.1f} us  {flops/(ms*1e-3)/1e12:7.2f} TF/s", flush=True)
def g(M, N, K, tA=False, tB=True):
    A = torch.randn((K, M) if tA else (M, K), device="cuda"); B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    bench(lambda: ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, tA, tB), 2.0*M*N*K, f"gemm {'T' if tA else 'N'}{'T' if tB else 'N'} M{M} N{N} K{K}")
g(4800, 48, 1024); g(4800, 64, 1024); g(4800, 1024, 64, False, False); g(4800, 1024, 48, False, False)
