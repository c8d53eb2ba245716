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
import os

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


# ---------------------------------------------------------------------------------------------------- SyncBN fast path
class RcclDirect:
    """RCCL called directly on the COMPUTE stream for the per-layer SyncBatchNorm exchanges (reference:
    nn.SyncBatchNorm.convert_sync_batchnorm, train_stage1.py:69 -- 55 all-gathers of [mean|invstd|var] in forward and 55
    all-reduces of the two backward sums per step, each a few KB).

    OPT-IN (TRIS_SYNCBN_COMM=rccl), because it measured SLOWER than torch.distributed: with a one-rank group
    (TRIS_FORCE_DIST=1, B = 48, same box, A/B/A/B) the step takes 52.0 ms without collectives, 55.5 ms through c10d and
    58.6-59.4 ms through this class.  The cost of these 110 tiny exchanges is not c10d's Work objects or stream hops but
    RCCL's own enqueue path (~55 us of host work per call), which a direct call pays just the same.  The way to take them
    off the step is to not call RCCL at all for them: tris_amd.comm.Mailbox (IPC-mapped peer buffers + flags).

    The same RCCL library that torch loaded (torch/lib/librccl.so -- no second copy in the process) is bound with ctypes
    and its own communicator is created once per process group (unique id from rank 0, distributed through
    torch.distributed); each exchange is one in-stream launch.  If the library or the communicator cannot be set up,
    `get()` returns None and the callers use torch.distributed."""

    _by_group = {}
    NCCL_FLOAT32, NCCL_SUM = 7, 0

    class _UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_ubyte * 128)]   # (c_ubyte: a c_char array reads back NUL-truncated)

    def __init__(self, group):
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = ctypes.CDLL(path)
        for name, args in (("ncclGetUniqueId", [ctypes.POINTER(self._UniqueId)]),
                           ("ncclCommInitRank", [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._UniqueId, ctypes.c_int]),
                           ("ncclAllGather", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                              ctypes.c_void_p]),
                           ("ncclAllReduce", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p]),
                           ("ncclCommDestroy", [ctypes.c_void_p])):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = args
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        self.lib = lib
        self.world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        uid = self._UniqueId()
        if rank == 0:
            self._chk(lib.ncclGetUniqueId(ctypes.byref(uid)))
        box = [ctypes.string_at(ctypes.byref(uid), 128) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        comm = ctypes.c_void_p()
        self._chk(lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, rank))   # binds the current HIP device
        self.comm = comm

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f"RCCL: {self.lib.ncclGetErrorString(rc).decode()} ({rc})")

    def all_gather_into(self, out, inp):
        self._chk(self.lib.ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), self.NCCL_FLOAT32, self.comm,
                                         torch.cuda.current_stream().cuda_stream))

    def all_reduce_sum(self, t):
        self._chk(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self.NCCL_FLOAT32, self.NCCL_SUM, self.comm,
                                         torch.cuda.current_stream().cuda_stream))

    @classmethod
    def get(cls, group=None):
        """the direct communicator of `group`, or None (then: torch.distributed).  Collective: every rank of the group
        must call it at the same point the first time (BatchNormFn does: first SyncBN forward of the first step)."""
        key = id(group) if group is not None else 0
        if key in cls._by_group:
            return cls._by_group[key]
        made = None
        if (backend(group) == "nccl" and os.environ.get("TRIS_SYNCBN_COMM", "c10d") == "rccl"
                and not torch.cuda.is_current_stream_capturing()):
            try:
                made = cls(group)
            except Exception as e:   # library missing / init refused: keep training on the c10d path, say so once
                import warnings
                warnings.warn(f"direct RCCL communicator for SyncBatchNorm unavailable ({e!r}); using torch.distributed")
                made = None
            # every rank must take the same path: agree (one tiny c10d collective, once)
            ok = torch.tensor([1 if made is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                made = None
        cls._by_group[key] = made
        return made

    @classmethod
    def reset(cls):
        """drop the cached communicators (call before destroying the process group: tests)"""
        for c in cls._by_group.values():
            if c is not None:
                try:
                    c.lib.ncclCommDestroy(c.comm)
                except Exception:
                    pass
        cls._by_group = {}


class Mailbox:
    """SyncBatchNorm exchanges through IPC-mapped peer mailboxes (csrc/comm.hip; include/tris_hip.h `tris_mbox_*`): one
    single-workgroup launch on the compute stream per exchange -- stores over xGMI into every peer's mailbox, per-sender
    flags, bounded spin on the own mailbox -- instead of an RCCL collective.  Default transport of the SyncBN statistics
    whenever every rank of the group can map every other rank's mailbox (one node); otherwise torch.distributed.

    Set-up is collective (first SyncBN forward of the first step): allocate, exchange the 64-byte IPC handles through
    torch.distributed, open the peers' handles, agree that everyone succeeded."""

    _by_group = {}
    CAP = 3 * 4096           # floats per sender block: [mean|invstd|var] of the widest BatchNorm (2048 channels) with headroom
    SPIN_LIMIT = int(os.environ.get("TRIS_MBOX_SPIN", "20000000"))  # polls (~ seconds) before an exchange gives up and raises the error flag

    def __init__(self, group):
        from . import _lib
        self.lib = _lib.load()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > _lib.CONSTS["TRIS_MBOX_MAX_WORLD"]:
            raise RuntimeError(f"world size {self.world} exceeds TRIS_MBOX_MAX_WORLD")
        own = ctypes.c_void_p()
        self._chk(self.lib.tris_mbox_alloc(ctypes.byref(own), self.CAP), "tris_mbox_alloc")
        self.own = own
        handle = ctypes.create_string_buffer(64)
        self._chk(self.lib.tris_mbox_ipc_handle(own, handle), "tris_mbox_ipc_handle")
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw), group=group)
        ptrs, self.opened = [], []
        for r, h in enumerate(handles):
            if r == self.rank:
                ptrs.append(own.value)
                continue
            p = ctypes.c_void_p()
            self._chk(self.lib.tris_mbox_ipc_open(ctypes.create_string_buffer(h, 64), ctypes.byref(p)), "tris_mbox_ipc_open")
            self.opened.append(p)
            ptrs.append(p.value)
        self.boxes = torch.tensor(ptrs, dtype=torch.int64, device="cuda")
        self.err = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.seq = 0

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with hipError_t {rc}")

    def exchange(self, src, out, mode, src1=None):
        """block = [src | src1]; mode 0: out[world][n] gathered, mode 1: out[n] = sum over ranks"""
        self.seq += 1
        self._chk(self.lib.tris_mbox_exchange_f32(src.data_ptr(), src.numel(), None if src1 is None else src1.data_ptr(),
                                                  0 if src1 is None else src1.numel(), out.data_ptr(), self.boxes.data_ptr(),
                                                  self.world, self.rank, self.seq, self.CAP, mode, self.SPIN_LIMIT,
                                                  self.err.data_ptr(), torch.cuda.current_stream().cuda_stream),
                  "tris_mbox_exchange_f32")

    def bn_combine(self, local_stats, C, count_per_rank, eps, momentum, stats, running_mean, running_var):
        """SyncBN forward exchange + combine in one launch (include/tris_hip.h: tris_mbox_bn_combine_f32)"""
        self.seq += 1
        self._chk(self.lib.tris_mbox_bn_combine_f32(local_stats.data_ptr(), C, count_per_rank, eps, momentum, stats.data_ptr(),
                                                    None if running_mean is None else running_mean.data_ptr(),
                                                    None if running_var is None else running_var.data_ptr(),
                                                    self.boxes.data_ptr(), self.world, self.rank, self.seq, self.CAP,
                                                    self.SPIN_LIMIT, self.err.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "tris_mbox_bn_combine_f32")

    def self_test(self):
        """One real exchange before the transport is trusted: every rank posts (rank + 1) * [1, 2, 3, 4] and must read back
        every peer's block within a short time-out (peer stores over xGMI that never become visible, a peer mapping that
        silently aliases local memory, ... -> exception -> the caller falls back to torch.distributed on EVERY rank)."""
        src = torch.arange(1, 5, device="cuda", dtype=torch.float32) * float(self.rank + 1)
        out = torch.zeros(self.world * 4, device="cuda", dtype=torch.float32)
        limit, self.SPIN_LIMIT = self.SPIN_LIMIT, min(self.SPIN_LIMIT, 4000000)
        try:
            self.exchange(src, out, 0)
            torch.cuda.synchronize()
        finally:
            self.SPIN_LIMIT = limit
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
        self.lib.tris_mbox_free(self.own)
        self.opened = []

    @classmethod
    def get(cls, group=None):
        key = id(group) if group is not None else 0
        if key in cls._by_group:
            return cls._by_group[key]
        made = None
        if os.environ.get("TRIS_SYNCBN_COMM", "mailbox") == "mailbox" and not torch.cuda.is_current_stream_capturing():
            try:
                made = cls(group)
                made.self_test()
            except Exception as e:
                import warnings
                warnings.warn(f"SyncBatchNorm mailboxes unavailable ({e!r}); using torch.distributed collectives")
                if made is not None:
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


def check_errors():
    """raise if any SyncBatchNorm mailbox exchange timed out (host sync: call where the loop syncs anyway)"""
    for m in Mailbox._by_group.values():
        if m is not None:
            m.check()


def shutdown():
    """release the SyncBN transports (call before dist.destroy_process_group)"""
    Mailbox.reset()
    RcclDirect.reset()


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
    c = RcclDirect.get(group) if inp.is_cuda else None
    if c is not None:
        c.all_gather_into(out, inp)
    else:
        all_gather_into(out, inp, group)


def syncbn_all_reduce_sum(t, group=None):
    if t.is_cuda and t.numel() <= Mailbox.CAP:
        m = Mailbox.get(group)
        if m is not None:
            m.exchange(t, t, 1)
            return
    c = RcclDirect.get(group) if t.is_cuda else None
    if c is not None:
        c.all_reduce_sum(t)
    else:
        all_reduce(t, group=group)
