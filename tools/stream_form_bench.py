"""Idle-device timing of the library's element-wise BatchNorm passes in their two loop forms (csrc/norm.hip: grid-stride with the
default cache policy vs one nontemporal piece per block; option STREAM_FORM).  usage: python tools/stream_form_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tris_amd.ops as o

def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

P, st = o.P, o._stream()
for M, C in ((307200, 256), (76800, 512), (307200, 64), (19200, 1024)):
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, r, dy, y, out, out2 = rnd(M, C), rnd(M, C), rnd(M, C), rnd(M, C), torch.empty(M, C, device="cuda"), torch.empty(M, C, device="cuda")
    mean, invstd, gamma, beta, s1, s2 = rnd(C), rnd(C).abs() + 0.5, rnd(C), rnd(C), rnd(C), rnd(C)
    mb = M * C * 4 / 1e6
    cases = {
        "bn_apply +resid (2R 1W)": (3, lambda: o.call("tris_bn_apply_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), P(r), P(out), M, C, 1, st)),
        "bn_apply (1R 1W)": (2, lambda: o.call("tris_bn_apply_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), None, P(out), M, C, 1, st)),
        "bn_bwd_apply mask from x (2R 1W)": (3, lambda: o.call("tris_bn_bwd_apply_f32", P(dy), None, P(x), P(mean), P(invstd), P(gamma), P(s1), P(s2), 1.0 / M, P(out), None, M, C, P(beta), st)),
        "bn_bwd_apply mask from y (3R 1W)": (4, lambda: o.call("tris_bn_bwd_apply_f32", P(dy), P(y), P(x), P(mean), P(invstd), P(gamma), P(s1), P(s2), 1.0 / M, P(out), None, M, C, None, st)),
        "bn_bwd_apply mask from y + dz (3R 2W)": (5, lambda: o.call("tris_bn_bwd_apply_f32", P(dy), P(y), P(x), P(mean), P(invstd), P(gamma), P(s1), P(s2), 1.0 / M, P(out), P(out2), M, C, None, st)),
    }
    print(f"M {M} C {C}: {mb:.0f} MB per tensor")
    for name, (streams, fn) in cases.items():
        res = []
        for sf in (0, 1):
            o.set_option("STREAM_FORM", sf if sf == 0 else 1)   # 1 = the default threshold (256 MB); forced below
            if sf: o.set_option("STREAM_FORM", 2)               # (2 MB: the streaming form for every shape here)
            us = timed(fn)
            res.append((us, streams * mb / us / 1e6 * 1e6 / 1e6))
        o.set_option("STREAM_FORM", None)
        print(f"   {name:40s} default form {res[0][0]:7.1f} us {streams * mb / res[0][0]:6.2f} TB/s | streaming form {res[1][0]:7.1f} us {streams * mb / res[1][0]:6.2f} TB/s")
