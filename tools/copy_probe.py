"""Dev tool: which torch-side ops launch device copies / elementwise kernels inside one training step."""
import os; os.environ.setdefault("TRIS_RANDOM_INIT", "1")  # synthetic weights (seed-fill)
import sys, warnings, torch
sys.path.insert(0, ".")
from torch.profiler import profile, ProfilerActivity
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.train_stage1 import freeze_aux, train_step
from tris_amd.utils.synth import seed_fill, synthetic_batch
args = get_parser().parse_args(["--backbone", "clip-RN50", "--size", "320", "--max_query_len", "20", "--negative_samples", "3", "--batch_size", "48"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = TRIS(args).cuda().train(); aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
seed_fill(model.state_dict(), 1234); seed_fill(aux.state_dict(), 4321); freeze_aux(aux)
bb, new = model.trainable_parameters()
opt = FusedAdamW([{"params": bb, "lr": 5e-6}, {"params": new}], lr=5e-5, weight_decay=0.01)
b = synthetic_batch(48, 320, 20, 3, seed=7)
img, ids, neg = b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda()
for _ in range(3): train_step(model, aux, opt, img, ids, neg, args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(model, aux, opt, img, ids, neg, args); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print(f"{e.key:28s} n={e.count:4d} dev_us={e.device_time_total:9.1f} shapes={str(e.input_shapes)[:90]}")
