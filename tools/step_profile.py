"""Dev tool: time the phases of one Stage-1 step at batch B (prints progressively)."""
import os, sys, time, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1")  # synthetic weights (seed-fill): no CLIP checkpoint needed
import torch
sys.path.insert(0, ".")
from tris_amd import ops
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.train_stage1 import freeze_aux, stage1_forward_losses
from tris_amd.utils.synth import seed_fill, synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print("cpus", os.cpu_count(), "B", B, flush=True)
args = get_parser().parse_args(["--size", "320", "--negative_samples", "3"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = TRIS(args).cuda().train()
    aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
seed_fill(m.state_dict(), 1234); seed_fill(aux.state_dict(), 4321); freeze_aux(aux)
bb, new = m.trainable_parameters()
opt = FusedAdamW([{"params": bb, "lr": 5e-6}, {"params": new}], lr=5e-5)
b = synthetic_batch(B, 320, 20, 3, seed=7)
img, ids, neg = b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = sync()
    losses, cls, sig = stage1_forward_losses(m, aux, img, ids, neg, args)
    t1 = sync()
    losses[0].backward()
    t2 = sync()
    opt.step()
    t3 = sync()
    print(f"iter {it}: fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  opt {1e3*(t3-t2):.1f} ms  loss {losses.tolist()}", flush=True)
ops._PROF_SHAPES = True
ops.profile_begin()
losses, cls, sig = stage1_forward_losses(m, aux, img, ids, neg, args)
losses[0].backward()
rec = ops.profile_end()
kinds = {}
for k, f, ms in rec:
    e = kinds.setdefault(k, [0, 0.0, 0.0]); e[0] += 1; e[1] += f; e[2] += ms
tot = sum(v[2] for v in kinds.values())
print(f"total GEMM-family ms {tot:.2f}")
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1][2])[:45]:
    print(f"  {v[2]:8.3f} ms {100*v[2]/tot:5.1f}%  x{v[0]:3d}  {v[1]/1e9:8.1f} GF  {v[1]/(v[2]*1e-3)/1e12:7.2f} TF/s  {k}", flush=True)
