"""Per-piece time stamps of the segmented captured step (TRIS_STEP_GRAPH=seg): where the graphs of the three streams start and end
within a step (HIP events behind every graph launch; averages over 5 steps).  usage: python tools/step_graph_marks.py"""
import os, sys, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1"); os.environ.setdefault("TRIS_GEMM_MODE", "h2"); os.environ["TRIS_STEP_GRAPH"] = "seg"   # (read once, at import of tris_amd.config)
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
for s in range(6):
    train_step(model, aux, opt, *bt, args, None)
g = model.__dict__["_tris_step_graph"][1]
acc = {}
for s in range(5):
    torch.cuda.synchronize()
    g.trace = True
    train_step(model, aux, opt, *bt, args, None)
    for n, t in g.marks():
        acc.setdefault(n, []).append(t)
g.trace = False
for n, v in acc.items():
    print(f"{n:16s} {sum(v)/len(v):8.3f} ms")
