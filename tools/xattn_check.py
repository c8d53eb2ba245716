"""Dev tool: the three cross-attention forward forms against fp64 at one shape, with timings (GPU box)."""
import math, sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
from tris_amd.config import cfg
B, P, N, C = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (48, 100, 48, 1024)
g = torch.Generator().manual_seed(0)
Qv, Kv, Vv = (torch.randn(B, P, C, generator=g).cuda() * 1.5 for _ in range(3))
Qt, Kt, Vt = (torch.randn(N, C, generator=g).cuda() * 1.5 for _ in range(3))
sc = 1.0 / math.sqrt(C)
Av = torch.softmax(Qv.double() @ Kt.double().t() * sc, dim=2)
At = torch.softmax(Qt.double() @ Kv.double().transpose(1, 2) * sc, dim=2)
rv, rl = Av @ Vt.double(), At @ Vv.double()
by = B * (4 * P * C + N * C) * 4 + 3 * N * C * 4
import os
if os.environ.get("PX_SLOTS"):
    ops.set_option("XATTN_PX_SLOTS", os.environ["PX_SLOTS"])
for form in ("pair", "slices", "px-x3", "px"):
    cfg.xattn_fused, cfg.xattn_px, cfg.xattn_h2 = form != "pair", form.startswith("px"), form == "px"
    with torch.no_grad():
        ops.profile_begin()
        for _ in range(3):
            nv, nl = ops.xattn(Qv, Kv, Vv, Qt, Kt, Vt)
        kinds = sorted({r[0] for r in ops.profile_end()})
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(50):
            ops.xattn(Qv, Kv, Vv, Qt, Kt, Vt)
        b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    ev = float((nv.double() - rv).abs().max() / rv.abs().max()); el = float((nl.double() - rl).abs().max() / rl.abs().max())
    if form.startswith("px"):
        kinds.append({0: "-", 1: "x3 pieces", 2: "h2 pieces"}[int(ops.query("tris_xattn_px_last_form"))])
    print(f"{form:7s} {kinds}  {us:6.1f} us  {by / us / 1e3:7.1f} GB/s  frac {by / us / 1e3 / 8000:.3f}   rel err new_vis {ev:.2e} new_lan {el:.2e}"
          f"  timed out: {ops.xattn_timed_out()}")

# backward on the saved probabilities: the pixel-row persistent launch (+ three split-K products) against the chain of batched products
cfg.xattn_fused, cfg.xattn_px, cfg.xattn_h2 = True, True, True
wv, wl = torch.randn(B, P, C, generator=g).cuda(), torch.randn(B, N, C, generator=g).cuda()
for form in ("chain", "px"):
    cfg.xattn_bwd_px = form == "px"
    q = [t.clone().requires_grad_(True) for t in (Qv, Kv, Vv, Qt, Kt, Vt)]
    times = []
    for it in range(12):
        nv, nl = ops.xattn(*q)
        loss = (nv * wv).sum() + (nl * wl).sum()
        for t in q:
            t.grad = None
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.profile_begin()
        a.record(); loss.backward(); b.record(); torch.cuda.synchronize()
        rec = ops.profile_end()
        kinds = sorted({r[0] for r in rec})
        dev = {k: round(sum(r[2] for r in rec if r[0] == k) * 1e3, 1) for k in kinds}
        times.append(a.elapsed_time(b) * 1e3)
    times = sorted(times[2:])
    print(f"backward {form:6s} device us by kind (HIP events around each call, {len(rec)} calls) {dev}  whole backward, host-issued: median {times[len(times) // 2]:7.1f} us (autograd of the two weighted sums included)  timed out: {ops.xattn_timed_out()}")
