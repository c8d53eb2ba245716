"""Which graphs of the segmented step run CONCURRENTLY with a graph of sleeping one-thread kernels on another stream, and which eager
kernels do?  (follow-up of tools/graph_step_overlap_probe.py)"""
import os, sys, time, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1"); os.environ["TRIS_STEP_GRAPH"] = "seg"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tris_amd import ops
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.train_stage1 import freeze_aux, train_step
from tris_amd.utils.synth import seed_fill, synthetic_batch
B = 48
args = get_parser().parse_args(["--size", "320", "--negative_samples", "3", "--max_query_len", "20"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = TRIS(args).cuda().train()
    aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
seed_fill(model.state_dict(), 1234); seed_fill(aux.state_dict(), 4321); freeze_aux(aux)
bb, new = model.trainable_parameters()
opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
b = synthetic_batch(B, 320, 20, 3, seed=7)
bt = (b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda())
for s in range(4):
    train_step(model, aux, opt, *bt, args, None)
torch.cuda.synchronize()
g = model.__dict__["_tris_step_graph"][1]
null, fresh = torch.cuda.current_stream(), torch.cuda.Stream()


def sleeper(n, stream, cyc=40000):
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        with torch.cuda.graph(gs, stream=stream):
            for _ in range(n):
                torch.cuda._sleep(cyc)
    return gs


def wall(fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def best(fns, n=3):
    return min(wall(fns) for _ in range(n))


S = sleeper(200, fresh)


def run_s():
    with torch.cuda.stream(fresh):
        S.replay()


ts = best([run_s])
print(f"sleeper graph alone {ts:.2f} ms")
items = [(f"fwd[{i}]", gr) for i, gr in enumerate(g.fwd)] + [(f"back[{i}]", t[0]) for i, t in enumerate(g.back)] + [("opt", g.g_opt)]
for name, gr in items:
    a = best([gr.replay])
    both = best([gr.replay, run_s])
    print(f"{name:10s} alone {a:6.2f} ms | + sleeper {both:6.2f} ms -> {'overlap' if both < max(a, ts) + 0.5 * min(a, ts) else 'SERIAL'}")
# the same for the first trunk graph launched on OTHER streams, the sleeper on each of eight fresh streams (hardware-queue classes)
own = ops.compute_stream()
streams = [torch.cuda.Stream() for _ in range(8)]
sleepers = [sleeper(200, st) for st in streams]
F0 = g.fwd[0]
for cname, cs in (("default", null), ("own", own), ("text", g.text)):
    def run_f0():
        with torch.cuda.stream(cs):
            F0.replay()
    a = best([run_f0])
    row = []
    for st, sg in zip(streams, sleepers):
        def run_sl():
            with torch.cuda.stream(st):
                sg.replay()
        row.append(best([run_f0, run_sl]))
    print(f"trunk part 1 on the {cname:8s} stream alone {a:5.2f} ms; + sleeper (3.5 ms) on 8 fresh streams: " + " ".join(f"{v:5.2f}" for v in row))
# and eager kernels on a non-default stream
# eager kernels of the trunk next to the sleeper: which kind does not share the device?
x = torch.randn(48 * 80 * 80, 256, device="cuda"); w = torch.randn(64, 256, device="cuda"); y = torch.empty(48 * 80 * 80, 64, device="cuda")
xs = torch.randn(48 * 160 * 160, 32, device="cuda")
mean, inv, ga, be = (torch.randn(256, device="cuda") for _ in range(4))
o = torch.empty_like(x)
cases = {
    "gemm 307200x64x256 (x3, eager)": lambda: ops.gemm(x, w, y, x.shape[0], 64, 256, 256, 256, 64, False, True),
    "bn_apply 307200x256": lambda: ops.call("tris_bn_apply_f32", ops.P(x), ops.P(mean), ops.P(inv), ops.P(ga), ops.P(be), None, ops.P(o), x.shape[0], 256, 1, ops._stream()),
    "elementwise relu 78M": lambda: torch.relu_(o),
}
for name, fn in cases.items():
    def rep():
        with torch.cuda.stream(own):
            for _ in range(20):
                fn()
    a = best([rep])
    both = best([rep, run_s])
    print(f"{name:32s} x20 alone {a:6.2f} ms | + sleeper {both:6.2f} ms -> {'overlap' if both < max(a, ts) + 0.5 * min(a, ts) else 'SERIAL'}")
