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
import ctypes


import torch
import torch.distributed as dist

from .config import cfg


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


# ---------------------------------------------------------------------------------------------------- SyncBN fast path
# (A direct in-stream ncclAllGather for these exchanges was built and measured in round 2 -- slower than torch.distributed,
# 58.6 vs 55.5 ms/step at one rank: the cost is RCCL's own enqueue path -- and removed in round 3; `git show 848321e:tris_amd/comm.py`.)
class Mailbox:
    """SyncBatchNorm exchanges through IPC-mapped peer mailboxes (csrc/comm.hip; include/tris_hip.h `tris_mbox_*`): one
    single-workgroup launch on the compute stream per exchange -- stores over xGMI into every peer's mailbox, per-sender
    flags, bounded spin on the own mailbox -- instead of an RCCL collective.  Default transport of the SyncBN statistics
    whenever every rank of the group can map every other rank's mailbox (one node); otherwise torch.distributed.

    Set-up is collective (first SyncBN forward of the first step): allocate, exchange the 64-byte IPC handles through
    torch.distributed, open the peers' handles, agree that everyone succeeded."""

    _by_group = {}
    CAP = 3 * 4096           # floats per sender block: [mean|invstd|var] of the widest BatchNorm (2048 channels) with headroom
    # polls before an exchange gives up, raises the error flag and poisons its outputs with NaN.  A poll is an uncached
    # system-scope load + s_sleep 8 (~1-2.5 us): the default 4e7 polls is about a minute -- a dead peer fails the step instead of
    # pinning a spinning workgroup on every surviving GPU for long (the slow-rank-0 case, checkpoint saves, is covered by the
    # barrier train_stage1.main puts behind them); cfg.mbox_spin / TRIS_MBOX_SPIN, read when an exchange is ISSUED (so that
    # cfg.override(mbox_spin=...) and later assignments count); an instance attribute `spin_limit` overrides it (self_test)
    spin_limit = None

    @property
    def SPIN_LIMIT(self):
        return int(self.spin_limit if self.spin_limit is not None else cfg.mbox_spin)

    def __init__(self, group):
        from . import _lib
        self.lib = _lib.load()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > _lib.CONSTS["TRIS_MBOX_MAX_WORLD"]:
            raise RuntimeError(f"world size {self.world} exceeds TRIS_MBOX_MAX_WORLD")
        # Set-up must be failure-SYMMETRIC: whatever goes wrong locally (allocation, IPC export), this rank still takes part in
        # the collective handle exchange -- contributing None -- so that its peers are not left inside all_gather_object
        # while it has moved on to the agreement all-reduce of get().  The decision is taken afterwards, from what everyone posted.
        self.own, self.opened = None, []
        raw, local_err = None, None
        try:
            own = ctypes.c_void_p()
            self._chk(self.lib.tris_mbox_alloc(ctypes.byref(own), self.CAP), "tris_mbox_alloc")
            self.own = own
            handle = ctypes.create_string_buffer(64)
            self._chk(self.lib.tris_mbox_ipc_handle(own, handle), "tris_mbox_ipc_handle")
            raw = bytes(handle.raw)
        except Exception as e:   # noqa: BLE001 -- reported below, after the collective
            local_err = e
        handles = [None] * self.world
        dist.all_gather_object(handles, raw, group=group)
        if local_err is not None:
            raise local_err
        if any(h is None for h in handles):
            raise RuntimeError(f"rank(s) {[r for r, h in enumerate(handles) if h is None]} could not export a mailbox")
        ptrs = []
        for r, h in enumerate(handles):
            if r == self.rank:
                ptrs.append(self.own.value)
                continue
            p = ctypes.c_void_p()
            self._chk(self.lib.tris_mbox_ipc_open(ctypes.create_string_buffer(h, 64), ctypes.byref(p)), "tris_mbox_ipc_open")
            self.opened.append(p)
            ptrs.append(p.value)
        self.boxes = torch.tensor(ptrs, dtype=torch.int64, device="cuda")
        self.err = torch.zeros(1, dtype=torch.int32, device="cuda")
        # the exchange counter lives on the device and is advanced by the kernels themselves: an exchange depends on no host
        # state, so SyncBatchNorm layers can be captured into the step's hipGraphs (tris_amd.graphs.SegmentedTrainStep)
        self.seq = torch.zeros(1, dtype=torch.int32, device="cuda")

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with hipError_t {rc}")

    def exchange(self, src, out, mode, src1=None):
        """block = [src | src1]; mode 0: out[world][n] gathered, mode 1: out[n] = sum over ranks"""
        self._chk(self.lib.tris_mbox_exchange_f32(src.data_ptr(), src.numel(), None if src1 is None else src1.data_ptr(),
                                                  0 if src1 is None else src1.numel(), out.data_ptr(), self.boxes.data_ptr(),
                                                  self.world, self.rank, self.seq.data_ptr(), self.CAP, mode, self.SPIN_LIMIT,
                                                  self.err.data_ptr(), torch.cuda.current_stream().cuda_stream),
                  "tris_mbox_exchange_f32")

    def bn_combine(self, local_stats, C, count_per_rank, eps, momentum, stats, running_mean, running_var):
        """SyncBN forward exchange + combine in one launch (include/tris_hip.h: tris_mbox_bn_combine_f32)"""
        self._chk(self.lib.tris_mbox_bn_combine_f32(local_stats.data_ptr(), C, count_per_rank, eps, momentum, stats.data_ptr(),
                                                    None if running_mean is None else running_mean.data_ptr(),
                                                    None if running_var is None else running_var.data_ptr(),
                                                    self.boxes.data_ptr(), self.world, self.rank, self.seq.data_ptr(), self.CAP,
                                                    self.SPIN_LIMIT, self.err.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "tris_mbox_bn_combine_f32")

    def bn_combine_bound(self, local_stats, C, count_per_rank, eps, momentum, stats, running_mean, running_var, gamma, beta, xhat_max,
                         resid_word, bound_word):
        """bn_combine + the bound word of the plane output in the same launch (tris_mbox_bn_combine_bound_f32)"""
        self._chk(self.lib.tris_mbox_bn_combine_bound_f32(
            local_stats.data_ptr(), C, count_per_rank, eps, momentum, stats.data_ptr(),
            None if running_mean is None else running_mean.data_ptr(), None if running_var is None else running_var.data_ptr(),
            self.boxes.data_ptr(), self.world, self.rank, self.seq.data_ptr(), self.CAP, self.SPIN_LIMIT, self.err.data_ptr(),
            gamma.data_ptr(), beta.data_ptr(), float(xhat_max), resid_word, bound_word, torch.cuda.current_stream().cuda_stream),
            "tris_mbox_bn_combine_bound_f32")

    def bn_bwd_exchange(self, sum_dz, sum_dzx, out, gamma, invstd, inv_count, xhat_max, dz_word, bound_word):
        """SyncBN backward: out[2C] = the two sums over all ranks + the bound word of dx (tris_mbox_bn_bwd_exchange_f32)"""
        self._chk(self.lib.tris_mbox_bn_bwd_exchange_f32(
            sum_dz.data_ptr(), sum_dzx.data_ptr(), sum_dz.numel(), out.data_ptr(), self.boxes.data_ptr(), self.world, self.rank,
            self.seq.data_ptr(), self.CAP, self.SPIN_LIMIT, self.err.data_ptr(), gamma.data_ptr(), invstd.data_ptr(),
            float(inv_count), float(xhat_max), dz_word, bound_word, torch.cuda.current_stream().cuda_stream),
            "tris_mbox_bn_bwd_exchange_f32")

    def self_test(self):
        """One real exchange before the transport is trusted: every rank posts (rank + 1) * [1, 2, 3, 4] and must read back
        every peer's block within a short time-out (peer stores over xGMI that never become visible, a peer mapping that
        silently aliases local memory, ... -> exception -> the caller falls back to torch.distributed on EVERY rank)."""
        src = torch.arange(1, 5, device="cuda", dtype=torch.float32) * float(self.rank + 1)
        out = torch.zeros(self.world * 4, device="cuda", dtype=torch.float32)
        self.spin_limit = min(self.SPIN_LIMIT, 4000000)
        try:
            self.exchange(src, out, 0)
            torch.cuda.synchronize()
        finally:
            self.spin_limit = None
        want = (torch.arange(1, self.world + 1, device="cuda", dtype=torch.float32)[:, None] *
                torch.arange(1, 5, device="cuda", dtype=torch.float32)[None, :]).reshape(-1)
        if int(self.err.item()) != 0 or not torch.equal(out, want):
            raise RuntimeError(f"mailbox self-test failed on rank {self.rank}: err={int(self.err.item())} got={out.tolist()}")

    def check(self):
        """host-side check of the time-out flag (synchronises): raise if an exchange was abandoned"""
        e = int(self.err.item())
        if e:
            raise RuntimeError(f"SyncBatchNorm mailbox exchange #{e} timed out on rank {self.rank}: a peer never posted its "
                               f"block (crashed rank, or ranks running different numbers of BatchNorm layers)")

    def close(self):
        for p in self.opened:
            self.lib.tris_mbox_ipc_close(p)
        if self.own is not None:
            self.lib.tris_mbox_free(self.own)
            self.own = None
        self.opened = []

    @classmethod
    def get(cls, group=None):
        key = id(group) if group is not None else 0
        if key in cls._by_group:
            return cls._by_group[key]
        made = None
        if cfg.syncbn_comm == "mailbox" and not torch.cuda.is_current_stream_capturing():
            made = cls.__new__(cls)
            try:
                made.__init__(group)     # (always reaches its collective handle exchange, whatever fails locally)
                made.self_test()
            except Exception as e:
                import warnings
                warnings.warn(f"SyncBatchNorm mailboxes unavailable ({e!r}); using torch.distributed collectives")
                try:
                    made.close()
                except Exception:
                    pass
                made = None
            ok = torch.tensor([1.0 if made is not None else 0.0], device="cuda")
            all_reduce(ok, op=dist.ReduceOp.MIN, group=group)    # every rank must take the same path
            if float(ok.item()) == 0.0:
                if made is not None:
                    made.close()
                made = None
        cls._by_group[key] = made
        return made

    @classmethod
    def reset(cls):
        for m in cls._by_group.values():
            if m is not None:
                torch.cuda.synchronize()
                m.close()
        cls._by_group = {}


def check_errors(collective=False):
    """Raise if any SyncBatchNorm mailbox exchange timed out.  Host sync: call where the loop syncs anyway -- on EVERY rank
    (a rank that gave up keeps NaN statistics for that layer, its peers may be the next to time out).  collective=True
    additionally takes the MAX of the flag over the group (one tiny all-reduce), so that every rank raises in the same
    step; call it at the same point on all ranks."""
    for m in Mailbox._by_group.values():
        if m is None:
            continue
        if collective and dist.is_initialized():
            flag = m.err.clone()
            all_reduce(flag, op=dist.ReduceOp.MAX, group=m.group)
            if int(flag.item()) != 0 and int(m.err.item()) == 0:
                raise RuntimeError(f"SyncBatchNorm mailbox exchange #{int(flag.item())} timed out on a peer of rank {m.rank}")
        m.check()


def shutdown():
    """release the SyncBN transports (call before dist.destroy_process_group)"""
    Mailbox.reset()


def syncbn_mailbox(group, numel):
    """the mailbox transport of `group` if it can carry a block of `numel` floats, else None"""
    if numel > Mailbox.CAP or not torch.cuda.is_available():
        return None
    return Mailbox.get(group)


def syncbn_all_gather_into(out, inp, group=None):
    if inp.is_cuda and inp.numel() <= Mailbox.CAP:
        m = Mailbox.get(group)
        if m is not None:
            m.exchange(inp, out, 0)
            return
    all_gather_into(out, inp, group)


def syncbn_all_reduce_sum(t, group=None):
    if t.is_cuda and t.numel() <= Mailbox.CAP:
        m = Mailbox.get(group)
        if m is not None:
            m.exchange(t, t, 1)   # (in place: the kernel's src / out parameters are not restrict-qualified)
            return
    all_reduce(t, group=group)
