import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
for rows, W in ((3840, 512), (960, 512), (19248, 768)):
    x = torch.randn(rows, W, device="cuda", requires_grad=True); g = torch.randn(W, device="cuda", requires_grad=True); b = torch.zeros(W, device="cuda", requires_grad=True)
    dy = torch.randn(rows, W, device="cuda")
    res = {}
    for blocks in (128, 512):
        ops.set_option("LN_BWD_BLOCKS", blocks)
        y = ops.layer_norm(x, g, b)
        for _ in range(3):
            x.grad = g.grad = b.grad = None; y.backward(dy, retain_graph=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            y.backward(dy, retain_graph=True)
        e1.record(); torch.cuda.synchronize()
        x.grad = g.grad = b.grad = None; y.backward(dy, retain_graph=True)
        res[blocks] = (e0.elapsed_time(e1) / 50 * 1e3, x.grad.clone(), g.grad.clone(), b.grad.clone())
    d = max(float((res[128][i] - res[512][i]).abs().max() / res[128][i].abs().max()) for i in (1, 2, 3))
    print(f"LN bwd rows {rows} W {W}: 128 blocks {res[128][0]:.1f} us | 512 blocks {res[512][0]:.1f} us | rel diff {d:.1e}")
