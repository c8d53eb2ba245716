"""Dev tool: Stage-1 inference latency at batch 1 (BASELINE.json configs[1]): eager launches vs one captured hipGraph."""
import os; os.environ.setdefault("TRIS_RANDOM_INIT", "1")  # synthetic weights (seed-fill)
import sys, time, warnings
import torch
sys.path.insert(0, ".")
from tris_amd.args import get_parser
from tris_amd.model.model_stage1 import TRIS
from tris_amd.utils.synth import seed_fill, synthetic_batch

args = get_parser().parse_args(["--size", "320"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = TRIS(args).cuda().eval()
seed_fill(m.state_dict(), 1234)
b = synthetic_batch(1, 320, 20, 0, seed=7)
img, ids = b["img"].cuda(), b["word_ids"].cuda()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ref = m(img, ids).clone()
    print(f"eager  full forward        : {timeit(lambda: m(img, ids)):8.3f} ms")
    vis = m.encode_visual(img)
    print(f"eager  cached-visual, text+fuse only: {timeit(lambda: m.forward_cached(vis, ids, 320)):8.3f} ms")
    g = torch.cuda.CUDAGraph()
    s_img, s_ids = img.clone(), ids.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): m(s_img, s_ids)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        s_out = m(s_img, s_ids)
    g.replay(); torch.cuda.synchronize()
    print("graph vs eager max diff:", float((s_out - ref).abs().max()))
    print(f"hipGraph full forward       : {timeit(lambda: g.replay()):8.3f} ms")
