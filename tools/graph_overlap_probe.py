"""Do two hipGraphs launched on two streams run CONCURRENTLY?  Graph A (N kernels of ~T us) is launched on stream a, graph B (same) on
stream b right behind it; HIP events give each graph's start and end.  serial: B starts when A ends; concurrent: B starts with A.
Swept over N (nodes per graph) and the number of graphs the same work is cut into.   usage: python tools/graph_overlap_probe.py"""
import time
import torch
dev = "cuda"
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
CYC = 40000          # ~20 us per kernel at ~2 GHz


def make(n, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n):
                torch.cuda._sleep(CYC)
    return g


def run(n, pieces):
    """n kernels per stream, as `pieces` graphs of n / pieces nodes each, submitted alternately a, b, a, b, ..."""
    ga = [make(n // pieces, sa) for _ in range(pieces)]
    gb = [make(n // pieces, sb) for _ in range(pieces)]
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e[0].record(sa)
        e[2].record(sb)
        for k in range(pieces):
            with torch.cuda.stream(sa):
                ga[k].replay()
            with torch.cuda.stream(sb):
                gb[k].replay()
        host = (time.perf_counter() - t0) * 1e3
        e[1].record(sa)
        e[3].record(sb)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        res.append((host, e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3]), wall))
    h, a, b, w = min(res, key=lambda r: r[3])
    one = a if pieces == 1 else None
    print(f"nodes/stream {n:4d} in {pieces:3d} graphs of {n // pieces:3d}: host {h:6.3f} ms  stream a {a:7.3f} ms  stream b {b:7.3f} ms  wall {w:7.3f} ms")


for n in (8, 32, 128, 256):
    for pieces in (1, 2, 4, 8, 16, 32):
        if n // pieces >= 2:
            run(n, pieces)
# reference: one stream alone
g = make(128, sa)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(sa)
with torch.cuda.stream(sa):
    g.replay()
e1.record(sa)
torch.cuda.synchronize()
print(f"one graph of 128 nodes alone: {e0.elapsed_time(e1):.3f} ms")
