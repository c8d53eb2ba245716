"""Dev tool: attention kernels (both implementations) at the shapes of the step and of a ViT-B/16 trunk."""
import os, sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
def bench(N, L, heads, causal, impl):
    os.environ["TRIS_MHA"] = impl
    W = heads * 64
    qkv = torch.randn(N, L, 3 * W, device="cuda", requires_grad=True)
    go = torch.randn(N, L, W, device="cuda")
    def fb():
        o = ops.mha(qkv, heads, causal); o.backward(go)
    for _ in range(3): fb()
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    it = 10
    tf = tb = 0.0
    for _ in range(it):
        a.record(); o = ops.mha(qkv, heads, causal); b.record(); o.backward(go); c.record(); torch.cuda.synchronize()
        tf += a.elapsed_time(b); tb += b.elapsed_time(c)
    fl = 4.0 * N * heads * L * L * 64 * (0.5 if causal else 1.0)
    print(f"{impl:5s} N{N} L{L} h{heads} causal={causal}: fwd {tf/it*1e3:8.1f} us ({fl/(tf/it)/1e9:6.2f} TF)  bwd {tb/it*1e3:8.1f} us ({3.5*fl/(tb/it)/1e9:6.2f} TF)", flush=True)
for impl in ("valu", "mfma"):
    bench(960, 20, 8, True, impl); bench(3840, 20, 8, True, impl); bench(48, 50, 12, False, impl)
bench(48, 401, 12, False, "mfma")
