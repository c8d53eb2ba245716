"""Which pairs of the segmented step's graphs overlap on the device?  Takes the captured step (tris_amd.graphs.SegmentedTrainStep) and
launches pairs (compute-stream graph, text-stream graph) with nothing else in flight: wall time of the pair against the two alone.
Variants: the compute graph on the process's default stream (what bench.py / the trainer do today) or on a stream of its own; the
text graph on the step's text stream or on a fresh stream.   usage: python tools/graph_step_overlap_probe.py"""
import os, sys, time, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1"); os.environ["TRIS_STEP_GRAPH"] = "seg"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
null = torch.cuda.current_stream()
own = torch.cuda.Stream()
fresh = torch.cuda.Stream()
print("current stream is the default stream:", null == torch.cuda.default_stream(), " text stream:", g.text, " wg stream:", g.wg)


def wall(pairs):
    """pairs: [(graph, stream), ...] launched in this order; -> ms until all are done"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for gr, st in pairs:
        with torch.cuda.stream(st):
            gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def best(pairs, n=3):
    return min(wall(pairs) for _ in range(n))


F0, FT, FA = g.fwd[0], g.g_ftext, g.g_faux
for name, cs in (("default stream", null), ("own stream", own)):
    for tname, ts in (("step's text stream", g.text), ("fresh stream", fresh), ("step's wgrad stream", g.wg)):
        a, t = best([(F0, cs)]), best([(FA, ts)])
        both = best([(F0, cs), (FA, ts)])
        rev = best([(FA, ts), (F0, cs)])
        print(f"trunk part 1 on {name:15s} {a:6.2f} ms | aux text tower on {tname:20s} {t:6.2f} ms | both {both:6.2f} ms (text first: {rev:6.2f})"
              f"  -> {'OVERLAP' if both < 0.8 * (a + t) else 'serial'}")


# is it the launch path or the device?  the same graphs next to a graph of sleeping one-thread kernels (no CUs, no memory traffic)
def sleeper(n, stream, cyc=40000):
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        with torch.cuda.graph(gs, stream=stream):
            for _ in range(n):
                torch.cuda._sleep(cyc)
    return gs


S_own, S_fresh = sleeper(200, own), sleeper(200, fresh)
s_own, s_fresh = best([(S_own, own)]), best([(S_fresh, fresh)])
print(f"sleeper graphs alone: {s_own:.2f} / {s_fresh:.2f} ms")
for nm, gr, st in (("trunk part 1 (default stream)", F0, null), ("aux text tower (text stream)", FA, g.text), ("TRIS text fwd (text stream)", FT, g.text)):
    a = best([(gr, st)])
    both = best([(gr, st), (S_fresh, fresh)])
    print(f"{nm:32s} {a:6.2f} ms | with a 200-node sleeper graph on another stream: {both:6.2f} ms (sleeper alone {s_fresh:.2f})")
# eager launches of the aux text tower on the text stream next to the trunk graph
ids_all = torch.cat([bt[1].long(), bt[2].long().reshape(B * 3, -1)], 0)


def eager_text():
    with torch.cuda.stream(g.text), torch.no_grad():
        aux.encode_text(ids_all)


def wall_fn(fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns:
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


eager_text(); torch.cuda.synchronize()
te = min(wall_fn([eager_text]) for _ in range(3))
tb = min(wall_fn([lambda: F0.replay(), eager_text]) for _ in range(3))
print(f"aux text tower EAGER on the text stream {te:.2f} ms | behind the trunk graph on the default stream: {tb:.2f} ms (trunk graph alone {best([(F0, null)]):.2f})")
