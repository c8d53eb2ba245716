"""Dev tool: one Stage-1 step with the distributed code path forced on a single rank (SyncBN collectives + segmented
all-reduce) must reproduce the plain single-GPU step."""
import os, sys, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1")  # synthetic weights (seed-fill): no CLIP checkpoint needed
import torch
import torch.distributed as dist
sys.path.insert(0, ".")
from tris_amd import ops
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.parallel import attach_reducer, convert_sync_batchnorm
from tris_amd.train_stage1 import freeze_aux, stage1_forward_losses
from tris_amd.utils.synth import seed_fill, synthetic_batch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
args = get_parser().parse_args(["--size", "320", "--negative_samples", "3"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = TRIS(args).cuda().train(); aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
seed_fill(aux.state_dict(), 4321); freeze_aux(aux)
bb, new = m.trainable_parameters()
opt = FusedAdamW([{"params": bb}, {"params": new}], lr=1e-5)
b = synthetic_batch(6, 320, 20, 3, seed=77)
img, ids, neg = b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda()
def run(sync):
    seed_fill(m.state_dict(), 5)
    for a in opt.arenas: a.g.zero_()
    red = None
    for mod in m.modules():
        if hasattr(mod, "process_group"): mod.process_group = None
    m.backbone.visual.grad_reducer = None
    m.backbone.grad_reducer = None
    if sync:
        convert_sync_batchnorm(m)
        red = attach_reducer(m, opt, force=True, check=True)
    losses, _, _ = stage1_forward_losses(m, aux, img, ids, neg, args)
    if red is not None: red.begin_step()
    losses[0].backward()
    if red is not None: red.finish()
    ops.wgrad_join(); torch.cuda.synchronize()
    return losses.clone(), [a.g.clone() for a in opt.arenas], {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
l0, g0, r0 = run(False)
l1, g1, r1 = run(True)
print("losses", l0.tolist(), l1.tolist())
for a, c in zip(g0, g1):
    print("grad arena max rel diff", float((a - c).abs().max()) / float(a.abs().max()))
print("running stats max diff", max(float((r0[k] - r1[k]).abs().max()) for k in r0))
named = dict(m.named_parameters())
worst = []
off = 0
for ai, ar in enumerate(opt.arenas):
    nm = {id(p): n for n, p in m.named_parameters()}
    for p, o in zip(ar.params, ar.offsets):
        d = float((g0[ai][o:o + p.numel()] - g1[ai][o:o + p.numel()]).abs().max()); s = float(g0[ai][o:o + p.numel()].abs().max()) + 1e-30
        worst.append((d / s, nm[id(p)]))
worst.sort(reverse=True)
worst = [w for w in worst if "attn_fusion.v_" not in w[1] or ".0.bias" not in w[1]]
import statistics
print("params", len(worst), "median rel diff", statistics.median(w[0] for w in worst), "n>1e-3:", sum(w[0] > 1e-3 for w in worst), "n>1e-5:", sum(w[0] > 1e-5 for w in worst))
for w in worst[:25]: print(f"  {w[0]:.3e} {w[1]}")
print("text/head params:", [(f"{w[0]:.1e}", w[1]) for w in worst if not w[1].startswith("backbone.visual")][:6])
__import__('tris_amd.comm', fromlist=['x']).shutdown()
dist.destroy_process_group()
