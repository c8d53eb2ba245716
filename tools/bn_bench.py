"""Per-shape timing of the three BatchNorm stream kernels (apply, backward reduce, backward apply) on the RN50 trunk's
own [M, C] shapes at 320 px, batch 48 -- achieved GB/s per kernel, to see which shapes fall short of the stream rate.
usage: python tools/bn_bench.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tris_amd._lib import call, query
from tris_amd.ops import P, workspace, _stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
SHAPES = [(160, 32), (160, 64), (80, 64), (80, 256), (80, 128), (40, 128), (40, 512), (40, 256), (20, 256), (20, 1024),
          (20, 512), (10, 512), (10, 2048)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'M':>9} {'C':>5} | apply us GB/s | reduce us GB/s | bwd-apply us GB/s")
for hw, C in SHAPES:
    M = B * hw * hw
    x = torch.randn(M, C, device="cuda")
    dy = torch.randn(M, C, device="cuda")
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    g = torch.rand(C, device="cuda") + 0.5
    b = torch.randn(C, device="cuda") * 0.1
    stats = torch.empty(3 * C, device="cuda")
    sums = torch.empty(2 * C, device="cuda")
    ws = workspace(query("tris_col_workspace_bytes", M, C))
    call("tris_bn_stats_f32", P(x), M, C, 1e-5, 0.1, P(stats), None, None, P(ws), _stream())
    mean, inv = stats[:C], stats[C:2 * C]
    ta = timeit(lambda: call("tris_bn_apply_f32", P(x), P(mean), P(inv), P(g), P(b), None, P(y), M, C, 1, _stream()))
    tr = timeit(lambda: call("tris_bn_bwd_reduce_f32", P(dy), None, P(x), P(mean), P(inv), M, C, P(sums), P(sums, C),
                             P(ws), P(g), P(b), None, _stream()))
    tb = timeit(lambda: call("tris_bn_bwd_apply_f32", P(dy), None, P(x), P(mean), P(inv), P(g), P(sums), P(sums, C),
                             1.0 / M, P(dx), None, M, C, P(b), _stream()))
    n = M * C * 4
    print(f"{M:>9} {C:>5} | {ta:7.1f} {2*n/ta/1e3:6.0f} | {tr:7.1f} {2*n/tr/1e3:6.0f} | {tb:7.1f} {3*n/tb/1e3:6.0f}")
