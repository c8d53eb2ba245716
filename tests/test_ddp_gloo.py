"""world_size-2 `gloo` test of the data-parallel host logic (runs on CPU): the gradient mean over ranks on the flat
arenas, weight/buffer broadcast at wrap time, rank-sharded synthetic data, SyncBN group plumbing and the cross-rank
(mean, var, count) combine formula used by the SyncBN kernel."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tris_amd.CLIP.clip.model import BatchNorm2d, Bottleneck
    from tris_amd.parallel import DataParallel, GradReducer, convert_sync_batchnorm
    from tris_amd.utils.synth import synthetic_batch
    torch.manual_seed(rank)  # different init per rank -> broadcast must equalise
    net = Bottleneck(16, 4, stride=2)
    wrapped = DataParallel(net)
    assert wrapped.module is net
    w = net.conv1.weight.detach().clone()
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    assert all(torch.equal(ws[0], x) for x in ws)
    convert_sync_batchnorm(net)
    assert all(m.process_group is not None for m in net.modules() if isinstance(m, BatchNorm2d))
    # gradient mean on two flat arenas, chunked
    flats = [torch.full((1000,), float(rank + 1)), torch.arange(10, dtype=torch.float32) * (rank + 1)]
    GradReducer(flats, chunk_mb=1).reduce()
    assert torch.allclose(flats[0], torch.full((1000,), 1.5))
    assert torch.allclose(flats[1], torch.arange(10, dtype=torch.float32) * 1.5)
    # segmented, backward-overlapped form: boundaries launch their segment when backward passes them
    flats = [torch.full((100,), float(rank + 1)), torch.full((50,), 10.0 * (rank + 1))]
    red = GradReducer(flats, chunk_mb=1)
    red.set_segments({"late": [(0, 0, 40), (1, 0, 50)], "early": [(0, 40, 100)]})
    x = torch.ones(3, requires_grad=True)
    y = red.boundary(red.boundary(x * 2, "early") * 3, "late")   # backward: "late" fires first, then "early"
    order = []
    orig = red._launch
    red._launch = lambda k: (order.append(k), orig(k))[1]
    y.sum().backward()
    assert order == ["late", "early"] and torch.allclose(x.grad, torch.full((3,), 6.0))
    red.finish()
    assert torch.allclose(flats[0], torch.full((100,), 1.5)) and torch.allclose(flats[1], torch.full((50,), 15.0))
    assert red.pending == [] and red.done == set()
    b = synthetic_batch(2, 8, 20, 3, seed=7, rank=rank)
    g = [torch.zeros_like(b["img"]) for _ in range(world)]
    dist.all_gather(g, b["img"])
    assert not torch.equal(g[0], g[1])
    # SyncBN combine (the formula of bn_sync_combine_kernel): per-rank (mean, biased var, count) -> statistics of the
    # concatenated batch, which is the N-rank correctness oracle of SURVEY.md 8(e)
    def data(r):  # equal per-rank counts, as DistributedSampler guarantees
        return torch.randn(6, 3, generator=torch.Generator().manual_seed(100 + r)) * (r + 1) + r
    x = data(rank)
    mine = torch.cat([x.mean(0), x.var(0, unbiased=False)])
    allv = [torch.zeros(6) for _ in range(world)]
    dist.all_gather(allv, mine)
    mean = sum(v[:3] for v in allv) / world
    var = sum(v[3:6] + (v[:3] - mean) ** 2 for v in allv) / world
    full = torch.cat([data(r) for r in range(world)], 0)
    assert torch.allclose(mean, full.mean(0), atol=1e-5) and torch.allclose(var, full.var(0, unbiased=False), atol=1e-4)
    if rank == 0:
        out.put((mean.numpy(), var.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    mean, var = out.get(timeout=10)
    assert np.isfinite(mean).all() and (var > 0).all()
