"""Dev tool: the fused cross-attention forward alone at the Stage-1 shape (B=48, P=100, N=48, C=1024)."""
import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
B, P, N, C = 48, 100, 48, 1024
g = torch.Generator().manual_seed(0)
Qv, Kv, Vv = (torch.randn(B, P, C, generator=g).cuda() for _ in range(3))
Qt, Kt, Vt = (torch.randn(N, C, generator=g).cuda() for _ in range(3))
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad():
    for _ in range(3):
        ops.xattn(Qv, Kv, Vv, Qt, Kt, Vt)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it):
        ops.xattn(Qv, Kv, Vv, Qt, Kt, Vt)
    b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / it * 1e3
by = B * (4 * P * C + N * C) * 4 + 3 * N * C * 4
print(f"xattn fwd {us:.1f} us  {by/us/1e3:.1f} GB/s  frac {by/us/1e3/8000:.3f}")
