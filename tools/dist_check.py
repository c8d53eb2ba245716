"""Preflight of the data-parallel Stage-1 path on N GPUs of one node (run BEFORE a scaling bench; reference:
DistributedDataParallel + SyncBatchNorm, /root/reference/train_stage1.py:69-70, 435-437):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29544 tools/dist_check.py

Every rank starts from DIFFERENT weights (the wrap-time broadcast must equalise them), trains `--steps` steps on its own shard
with the order check on (NaN-poisoned gradient arenas) and asserts:
  * the SyncBatchNorm statistics travel through the IPC peer mailboxes (unless TRIS_SYNCBN_COMM says otherwise);
  * every reducer segment is released, the trunk stages from inside backward in completion order;
  * the token-embedding gradient went through the sparse (ids, rows) exchange;
  * parameters AND optimiser moments are bit-identical on all ranks after the last step;
  * losses are finite and no mailbox exchange timed out on any rank (comm.check_errors(collective=True)).
Prints one JSON line on rank 0 with the transport, the exposed communication time and the step time.
`--single` runs the same checks in ONE process with the collectives forced (code-path check).  TRIS_STEP_GRAPH=seg runs the steps
through the segmented hipGraph replay (collectives issued between graph replays), TRIS_GEMM_MODE=h2 in the h2 arithmetic."""
import argparse
import json
import os
import sys
import time
import warnings

os.environ.setdefault("TRIS_RANDOM_INIT", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8, help="images per rank")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--single", action="store_true")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    force = a.single or world == 1
    from tris_amd import comm, ops
    from tris_amd.args import get_parser
    from tris_amd.CLIP import clip
    from tris_amd.model.model_stage1 import TRIS
    from tris_amd.optim import FusedAdamW
    from tris_amd.parallel import DataParallel, attach_reducer, convert_sync_batchnorm
    from tris_amd.train_stage1 import freeze_aux, train_step
    from tris_amd.utils.synth import seed_fill, synthetic_batch
    cs = ops.place_streams()
    if cs is not None:
        torch.cuda.set_stream(cs)
    args = get_parser().parse_args(["--size", "320", "--max_query_len", "20", "--negative_samples", "3", "--batch_size", str(a.batch)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = TRIS(args).cuda().train()
        aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
    seed_fill(net.state_dict(), 1234 + 17 * rank)      # different per rank on purpose
    seed_fill(aux.state_dict(), 4321)
    freeze_aux(aux)
    convert_sync_batchnorm(net)
    model = DataParallel(net)
    bb, new = net.trainable_parameters()
    opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
    red = attach_reducer(model, opt, force=force, check=True)
    assert red.active
    t_step = None
    for s in range(a.steps):
        b = synthetic_batch(a.batch, 320, 20, 3, seed=100 + s, rank=rank)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = train_step(model, aux, opt, b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda(), args, reducer=red)
        torch.cuda.synchronize()
        t_step = time.perf_counter() - t0
        assert bool(torch.isfinite(losses).all()), (rank, s, losses.tolist())
        log = list(red.launch_log)
        want = ["heads", "embed", "text_hi", "text_mid", "text", "layer4", "layer3", "layer2", "layer1", "stem"]
        assert sorted(log) == sorted(want), (rank, s, log)
        assert [k for k in log if k in ("heads", "layer4", "layer3", "layer2", "layer1")] == ["heads", "layer4", "layer3", "layer2", "layer1"], log
        if red.sparse_embed:
            assert len(red.sparse_log) == 1 and red.sparse_log[0][0] == world * a.batch * 20, (rank, s, red.sparse_log)
    comm.check_errors(collective=True)
    transport = "mailbox" if any(m is not None for m in comm.Mailbox._by_group.values()) else "torch.distributed"
    from tris_amd.config import cfg
    if cfg.syncbn_comm == "mailbox":
        assert transport == "mailbox", f"rank {rank}: SyncBatchNorm fell back to {transport}"
    # replicas: parameters and both Adam moments bit-identical everywhere (compared through an exact integer checksum)
    sig = []
    for ar in opt.arenas:
        for t in (ar.p, ar.m, ar.v):
            v = t.view(torch.int32).to(torch.int64)
            sig += [v.sum(), (v * torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003).sum()]
    sig = torch.stack(sig)
    all_sig = [torch.empty_like(sig) for _ in range(world)]
    dist.all_gather(all_sig, sig)
    assert all(torch.equal(all_sig[0], x) for x in all_sig), f"rank {rank}: replicas diverged"
    exposed = red.exposed_ms()
    if rank == 0:
        print(json.dumps({"dist_check": "ok", "world": world, "forced_single": force, "per_rank_batch": a.batch, "steps": a.steps,
                          "step_issue": cfg.step_graph, "replayed": "_tris_step_graph" in net.__dict__, "gemm_mode": ops.get_gemm_mode(),
                          "sync_bn_transport": transport, "sparse_embed_rows": red.sparse_log[0] if red.sparse_log else None,
                          "comm_exposed_ms_last_step": None if exposed is None else round(exposed, 3),
                          "ms_last_step": round(t_step * 1e3, 2), "losses_last_step": [round(v, 5) for v in losses.tolist()]}))
    dist.barrier()
    comm.shutdown()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
