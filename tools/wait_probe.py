"""How long does each stream WAIT on another one inside a training step, untraced?  Every torch.cuda.Stream.wait_stream /
wait_event of the step is bracketed by two events recorded on the WAITING stream (nothing else is enqueued between them, so
their distance is the idle time the dependency costs that stream).  Eager three-stream step, B = 48, h2 unless TRIS_GEMM_MODE says
otherwise.  usage: python tools/wait_probe.py [steps=5]"""
import os, sys, traceback, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1"); os.environ.setdefault("TRIS_GEMM_MODE", "h2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.train_stage1 import freeze_aux, train_step
from tris_amd.utils.synth import seed_fill, synthetic_batch
from tris_amd import ops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 48
args = get_parser().parse_args(["--size", "320", "--negative_samples", "3", "--max_query_len", "20"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = TRIS(args).cuda().train()
    aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
seed_fill(model.state_dict(), 1234); seed_fill(aux.state_dict(), 4321); freeze_aux(aux)
cs = ops.place_streams()
if cs is not None:
    torch.cuda.set_stream(cs)
bb, new = model.trainable_parameters()
opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
b = synthetic_batch(B, 320, 20, 3, seed=7)
bt = (b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda())
for s in range(8):
    train_step(model, aux, opt, *bt, args, None)
torch.cuda.synchronize()
names = {}
def sname(st):
    return names.setdefault(st.cuda_stream, f"stream{len(names)}")
sname(torch.cuda.current_stream())
log = []
orig_ws, orig_we = torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event
def where():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "tris_amd" in f.filename and "wait_probe" not in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno}"
    return "?"
def ws(self, other):
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(self); orig_ws(self, other); c.record(self)
    log.append((sname(self), "wait_stream", where(), a, c))
def we(self, ev):
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(self); orig_we(self, ev); c.record(self)
    log.append((sname(self), "wait_event", where(), a, c))
torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = ws, we
acc, order, tot = {}, [], []
for s in range(steps):
    log.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    train_step(model, aux, opt, *bt, args, None)
    e1.record()
    torch.cuda.synchronize()
    tot.append(e0.elapsed_time(e1))
    seen = {}
    for st, kind, w, a, c in log:
        k = (st, kind, w); seen[k] = seen.get(k, 0) + 1; k = k + (seen[k],)
        if k not in acc:
            acc[k] = []; order.append(k)
        acc[k].append((e0.elapsed_time(a), a.elapsed_time(c)))
torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = orig_ws, orig_we
print(f"step (one step between two synchronisations, so the host starts level with the GPU): {sum(tot)/len(tot):.2f} ms; streams: {names}")
print("  at ms   waited ms   stream   kind         where (n-th at that line)")
mainw = 0.0
for k in order:
    at = sum(v[0] for v in acc[k]) / len(acc[k]); w = sum(v[1] for v in acc[k]) / len(acc[k])
    if k[0] == "stream0": mainw += w
    if w >= 0.02:
        print(f"{at:8.2f} {w:10.3f}   {k[0]:8s} {k[1]:12s} {k[2]} #{k[3]}")
print(f"main stream waited {mainw:.2f} ms per step in total")
