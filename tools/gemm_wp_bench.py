"""Dev tool: the Linear / 1x1 products with pre-split weight planes (B by LDS-DMA, csrc/gemm_fast.h B_NK_PRE) against the plain x3
kernel (both operands split in-kernel), same shapes, same process, autotuned tiles; also checks that the results agree."""
import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
from tris_amd.planes import WeightPlanes
torch.manual_seed(0)
def tm(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
def one(M, N, K, name=""):
    x = torch.randn(M, K, device="cuda"); w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05)
    dy = torch.randn(M, N, device="cuda")
    wp = WeightPlanes([("w", w)]); wp.refresh()
    with torch.no_grad():
        y0 = ops.linear(x, w)
        with WeightPlanes.active():
            y1 = ops.linear(x, w)
        err = float((y0 - y1).abs().max()) / float(y0.abs().max())
        t0 = tm(lambda: ops.linear(x, w))
        with WeightPlanes.active():
            t1 = tm(lambda: ops.linear(x, w))
    fl = 2.0 * M * N * K
    print(f"{name:10s} fwd  M{M:7d} N{N:5d} K{K:5d}  plain {t0*1e3:8.1f} us {fl/t0/1e9:7.1f} TF  planes {t1*1e3:8.1f} us {fl/t1/1e9:7.1f} TF  x{t0/t1:.2f}  relerr {err:.1e}", flush=True)
    # data gradient: dx = dy . W  (planes: transposed planes turn it into the same NT product)
    dx0 = torch.empty(M, K, device="cuda"); dx1 = torch.empty(M, K, device="cuda")
    ws = ops.workspace(0)
    f0 = lambda: ops.gemm(dy, w, dx0, M, K, N, N, K, K, False, False)
    p = w._tris_wp
    f1 = lambda: ops._wp_call("tris_gemm_wp_f32", ops.P(dy), p[2], p[3], ops.P(dx1), M, K, N, None, None, 0, ops.P(ws), ws.numel() * 4, None, None, ops._stream())
    f0(); assert f1()
    err = float((dx0 - dx1).abs().max()) / float(dx0.abs().max())
    t0, t1 = tm(f0), tm(f1)
    print(f"{name:10s} dgrd M{M:7d} N{K:5d} K{N:5d}  plain {t0*1e3:8.1f} us {fl/t0/1e9:7.1f} TF  planes {t1*1e3:8.1f} us {fl/t1/1e9:7.1f} TF  x{t0/t1:.2f}  relerr {err:.1e}", flush=True)
for M, N, K, nm in [(4096, 4096, 4096, "big"), (2400, 3072, 768, "vit fc"), (2400, 768, 3072, "vit proj"), (2400, 2304, 768, "vit qkv"),
                    (2400, 768, 768, "vit out"), (3840, 2048, 512, "auxtxt fc"), (3840, 512, 2048, "auxtxt pj"), (960, 2048, 512, "txt fc"),
                    (19200, 1024, 256, "l3 conv3"), (19200, 256, 1024, "l3 conv1"), (76800, 512, 128, "l2 conv3"), (76800, 128, 512, "l2 conv1"),
                    (307200, 256, 64, "l1 conv3"), (307200, 64, 256, "l1 conv1"), (4800, 2048, 512, "l4 conv3"), (4800, 512, 2048, "l4 conv1"),
                    (4800, 1024, 2048, "visproj")]:
    one(M, N, K, nm)
