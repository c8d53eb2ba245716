"""Dev tool: the split-K weight-gradient products whose outputs are small, with and without the wide slab reduce (option
REDUCE_WIDE), autotuned on an idle device (GPU box):  python tools/reduce_bench.py"""
import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
torch.manual_seed(0)
SHAPES = [(64, 64, 307200), (64, 256, 307200), (256, 64, 307200), (128, 256, 307200), (128, 512, 76800), (512, 128, 76800),
          (256, 512, 76800), (512, 256, 76800), (48, 1024, 19200), (48, 1024, 4800), (256, 1024, 19200), (1024, 256, 19200)]
def run(M, N, K):
    A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); C = torch.empty(M, N, device="cuda")
    f = lambda: ops.gemm(A, B, C, M, N, K, M, N, N, True, False)
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ref = (A[:4096].double().t() @ B[:4096].double())
    return a.elapsed_time(b) / 20 * 1e3, C.clone()
import os
for mode in ("x3", "h2"):
    ops.set_gemm_mode(mode)
    for M, N, K in SHAPES:
        us, _ = run(M, N, K)
        print(f"REDUCE_WIDE={os.environ.get('TRIS_REDUCE_WIDE', '1')} {mode} wgrad M{M:5d} N{N:5d} K{K:6d}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s", flush=True)
