"""Collective plumbing of the data-parallel path (reference: DistributedDataParallel + SyncBatchNorm over NCCL,
train_stage1.py:69-70, 435-437).

Two exchange steps exist on the Stage-1 path: the gradient mean over the flat arenas and the per-layer
SyncBatchNorm statistics.  Both go through the helpers below so that one place decides HOW a collective is carried:

  * backend "nccl" (= RCCL over xGMI, the production path): torch.distributed on the device tensors as they are;
  * backend "gloo" (CPU tests of the host logic, and the two-ranks-on-one-GPU parity test of the real model: RCCL
    refuses two ranks on one device): the payload is staged through host memory around the collective.  That keeps
    the code under test -- reducer ordering, SyncBN math, scaling by the world size -- identical to production while
    only the wire differs.
"""
import torch
import torch.distributed as dist


def backend(group=None):
    return dist.get_backend(group) if dist.is_initialized() else None


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


class _Done:
    """handle of a collective that completed synchronously (host-staged gloo path)"""

    def wait(self):
        return True


def all_reduce(t, op=None, group=None, async_op=False):
    """in-place all-reduce of a contiguous tensor; returns a handle with .wait() when async_op"""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and backend(group) == "gloo":
        h = t.detach().cpu()            # (synchronises with the producing stream)
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


def all_gather_into(out, inp, group=None):
    """out [world * n] <- concat over ranks of inp [n]"""
    if inp.is_cuda and backend(group) == "gloo":
        h = inp.detach().cpu()
        o = torch.empty(out.numel(), dtype=h.dtype)
        dist.all_gather_into_tensor(o, h.reshape(-1), group=group)
        out.copy_(o.view_as(out))
        return
    dist.all_gather_into_tensor(out, inp, group=group)


def broadcast(t, src=0, group=None):
    if t.is_cuda and backend(group) == "gloo":
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src=src, group=group)
