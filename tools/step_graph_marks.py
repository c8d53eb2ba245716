"""Per-piece time stamps of the segmented captured step (TRIS_STEP_GRAPH=seg): where the graphs of the three streams start and end
within a step (HIP events behind every graph launch; averages over 5 steps).  usage: python tools/step_graph_marks.py [--dist]
--dist: the data-parallel code path forced at world size 1 (RCCL process group, SyncBatchNorm through the mailbox, gradient reducer
with its collectives issued between the replayed graphs): where its overhead over the plain step sits."""
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
DIST = "--dist" in sys.argv
bb, new = model.trainable_parameters()
opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
reducer = None
if DIST:
    import torch.distributed as dist
    from tris_amd.parallel import attach_reducer, convert_sync_batchnorm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from tris_amd import ops as _ops
    cs = _ops.place_streams()        # (as train_stage1.main does: the streams on hardware queues the collective backend is not on)
    if cs is not None:
        torch.cuda.set_stream(cs)
    convert_sync_batchnorm(model)
    reducer = attach_reducer(model, opt, force=True)
b = synthetic_batch(B, 320, 20, 3, seed=7)
bt = (b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda())
for s in range(6):
    train_step(model, aux, opt, *bt, args, None, reducer)
g = model.__dict__["_tris_step_graph"][1]
acc = {}
for s in range(5):
    torch.cuda.synchronize()
    g.trace = True
    train_step(model, aux, opt, *bt, args, None, reducer)
    for n, t in g.marks():
        acc.setdefault(n, []).append(t)
g.trace = False
for n, v in acc.items():
    print(f"{n:16s} {sum(v)/len(v):8.3f} ms")
if DIST:
    import time
    for name, form in (("replayed", "seg"), ("eager", "0")):
        from tris_amd.config import cfg
        cfg.step_graph = form
        for _ in range(3):
            train_step(model, aux, opt, *bt, args, None, reducer)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            train_step(model, aux, opt, *bt, args, None, reducer)
        torch.cuda.synchronize()
        print(f"dist path, one rank, {name}: {(time.perf_counter() - t0) * 100:.3f} ms/step")
    dist.destroy_process_group()
