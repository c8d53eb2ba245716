"""Dev tool: is the B=48 step launch-bound?  Host time to ENQUEUE a step vs device time to finish it."""
import os, sys, time, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1")  # synthetic weights (seed-fill): no CLIP checkpoint needed
import torch
sys.path.insert(0, ".")
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.train_stage1 import freeze_aux, train_step
from tris_amd.utils.synth import seed_fill, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
args = get_parser().parse_args(["--size", "320", "--negative_samples", "3"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = TRIS(args).cuda().train(); aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
seed_fill(m.state_dict(), 1234); seed_fill(aux.state_dict(), 4321); freeze_aux(aux)
bb, new = m.trainable_parameters()
opt = FusedAdamW([{"params": bb, "lr": 5e-6}, {"params": new}], lr=5e-5)
b = synthetic_batch(B, 320, 20, 3, seed=7)
img, ids, neg = b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda()
for _ in range(3): train_step(m, aux, opt, img, ids, neg, args)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_step(m, aux, opt, img, ids, neg, args)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"B={B}: host enqueue {sorted(enq)[len(enq)//2]:.1f} ms, step (synced) {sorted(tot)[len(tot)//2]:.1f} ms")
