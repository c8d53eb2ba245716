"""Host-side operators of the TRIS Stage-1 hot path: thin autograd wrappers over the C ABI (include/tris_hip.h).

PyTorch supplies device memory, the current HIP stream and the autograd tape -- plumbing.  Every
numeric op below is a kernel from libtris_hip.so; there is no eager / CPU fallback and CPU tensors
are rejected loudly.

Layout conventions: activations are channels-last, i.e. an image tensor is a contiguous
[B, H, W, C] array; conv weights keep the reference's [Cout, Cin, kh, kw] *shape* in
torch.channels_last memory so state dicts stay compatible.

Weight gradients: a parameter may carry a "gradient sink" (`p.grad` is a view into a flat arena,
see tris_amd.optim).  Backward kernels then write the weight gradient straight into that arena
(each parameter is used once per step) and return None to autograd.
"""
import math
import os

import torch

from . import _lib
from ._lib import CONSTS, call, query
from .config import cfg

EW = CONSTS


class NoGpuError(RuntimeError):
    pass


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """raw handle of the current HIP stream (the launch path calls this ~1500 times per step: the raw accessor avoids building a
    torch.cuda.Stream object each time)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise NoGpuError("tris_amd ops run on MI355X only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32 and t.dtype != torch.int64 and t.dtype != torch.uint8:
            raise TypeError(f"unsupported dtype {t.dtype}")


def P(t, off=0):
    return None if t is None else t.data_ptr() + 4 * off


_WS = {}


def workspace(nbytes):
    """Scratch arena per (device, stream): kernels on one stream run in order, so serial reuse is safe; work issued on
    a second stream (the text encoders overlap the RN50 trunk, tris_amd.model.model_stage1) gets its own arena."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    cur = _WS.get(key)
    if cur is None or cur.numel() * 4 < nbytes:
        n = max(int(nbytes), 256 << 20)
        cur = torch.empty(n // 4 + 1, dtype=torch.float32, device="cuda")
        _WS[key] = cur
    return cur


_TICKETS = {}
TICKET_SLOTS = 1 << 14


def splitk_tickets():
    """ticket array of the fused split-K finish (tris_splitk_tickets_next) for the current (device, stream): zero when created, left
    zero by every launch; products launched on one stream run in order, so they share it"""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    cur = _TICKETS.get(key)
    if cur is None:
        if torch.cuda.is_current_stream_capturing():
            return None        # (never allocated under capture: the product then takes the two-launch form)
        cur = _TICKETS[key] = torch.zeros(TICKET_SLOTS, dtype=torch.int32, device="cuda")
    return cur


# ---- arithmetic of the dense products --------------------------------------------------------------------------------------
# 'h2' (default since round 5 -- the arithmetic bench.py reports and the trainer runs are one and the same): two fp16 pieces per
#   operand (the residual pre-scaled by 2^11), three f16 MFMAs per product, one power-of-two scale per TENSOR formed on the device from
#   its largest magnitude: fp32-class accuracy over a 2^27 dynamic range inside a tensor, half the matrix work of x3
#   (csrc/x3_split.h, "h2 products" below).  A product whose operands have no amax runs in x3.
# 'x3': split-bf16 -- three bf16 pieces per fp32 operand, six bf16 MFMAs per product, fp32 accumulate: fp32-class accuracy at ~2.6x
#   the f32-MFMA rate; also what the kernel library itself runs when a product is not armed for h2.
# 'f32': the f32-input MFMA (bit-equal to an fmaf chain).
_ARITH = "h2"     # (TRIS_GEMM_MODE at import overrides; the library's own default for unarmed products is x3 either way)


def set_gemm_mode(mode):
    """'x3' | 'h2' | 'f32' (env TRIS_GEMM_MODE at import).  h2 is per product on top of x3: the library's own default
    arithmetic (tris_set_gemm_mode) stays x3 for whatever cannot run in h2."""
    global _ARITH
    assert mode in ("f32", "x3", "h2"), mode
    call("tris_set_gemm_mode", 0 if mode == "f32" else 1)
    _ARITH = mode


def get_gemm_mode():
    return _ARITH


_TLS = __import__("threading").local()


def _set_thread_mode(mode):
    """per-thread arithmetic override of the dense products (None = none; 'f32' | 'x3'); returns the previous override"""
    prev = getattr(_TLS, "mode", None)
    _TLS.mode = mode
    call("tris_set_gemm_mode_thread", -1 if mode is None else {"f32": 0, "x3": 1}[mode])
    return prev


def _thread_mode_is_set():
    return getattr(_TLS, "mode", None) is not None


def set_autotune(on):
    """per-shape (tile, split-K) autotuning of the GEMM core on/off (include/tris_hip.h: tris_set_autotune)"""
    call("tris_set_autotune", int(bool(on)))


def set_option(name, value):
    """developer option of the kernel library (include/tris_hip.h tris_set_option): FORCE_TILE, FORCE_PIPE, CONV_DIRECT,
    WGRAD_DIRECT, BN_FOLD, TUNE_LOG, ...; value None restores the default"""
    call("tris_set_option", name.encode(), None if value is None else str(value).encode())


class option:
    """`with ops.option("CONV_DIRECT", 0): ...` -- a library option for a scope (tests)"""

    def __init__(self, name, value, restore=None):
        self.name, self.value, self.restore = name, value, restore

    def __enter__(self):
        set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.restore)
        return False


if os.environ.get("TRIS_GEMM_MODE"):
    set_gemm_mode(os.environ["TRIS_GEMM_MODE"])


def cl_weight(w):
    """Conv weight in the kernel layout [Cout][kh][kw][Cin] (= channels_last memory)."""
    if w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
        return w.contiguous(memory_format=torch.channels_last)
    return w


# ----------------------------------------------------------------------------------------------- weight-gradient stream
# Weight gradients are consumed only by the optimiser (or the data-parallel reducer), never by the rest of backward.
# They are issued on their own HIP stream so a wgrad (split-K, often HBM-bound in the early layers) overlaps the dgrad /
# BatchNorm-backward chain that continues on the main stream.  `wgrad_join()` makes the current stream wait for them.
_WG = {}


# Side streams under stream capture: a capture that forks (event record on the capturing stream, wait on the side stream) and
# joins again before it ends records the side streams' launches as parallel branches of the graph.  The small inference graphs
# (tris_amd.graphs.GraphedStage1Eval) stay single-stream; the captured training step switches this on around its capture.
_CAPTURE_STREAMS = False


def streams_allowed():
    """may work be forked onto a side stream here?  (not under a stream capture, unless the capture asked for it)"""
    return _CAPTURE_STREAMS or not torch.cuda.is_current_stream_capturing()


# Segmented capture (tris_amd.graphs.SegmentedTrainStep): the step is captured as a chain of SINGLE-STREAM hipGraphs (the form
# the ROCm runtime launches through its packet path), cut at autograd boundaries; weight-gradient work is not forked onto its
# stream from inside a capture but handed to the capturing object, which records it as a graph of its own for that stream.
_SEG = None


def cut(x):
    """Autograd segment boundary.  A no-op (returns x) except while a segmented capture is being recorded: there the returned
    tensor is a fresh leaf on x's storage, and the capture drives the backward segment by segment (x <- leaf.grad)."""
    if _SEG is None or not x.requires_grad or not torch.is_grad_enabled():
        return x
    return _SEG.cut(x)


def _wgrad_enabled():
    if _SEG is not None:
        return True
    return cfg.wgrad_stream and streams_allowed()


# ---- side streams on their own hardware queues ------------------------------------------------------------------------------
# The HIP runtime maps streams onto a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) round-robin in creation
# order, and two streams on one queue run one after the other.  Which queue a new stream gets depends on how many streams the
# process created before -- a collective backend that creates three or four of its own is enough to put the text or the
# weight-gradient stream on the COMPUTE stream's queue, and the overlap this path is built on silently disappears (measured:
# +3.7 ms per step as soon as an RCCL communicator exists, reproduced without RCCL by creating three dummy streams first;
# DESIGN.md "Streams").  So the side streams are not taken as they come: a handful of candidates is created and PROBED -- a
# device-side sleep on one stream, a tiny kernel on the other, HIP-event timestamps tell whether they overlapped -- and the
# first ones that run concurrently with the compute stream and with each other are kept.
_CAL = {}


def _calibrated_streams(dev):
    """up to three streams of device `dev` that the hardware runs concurrently with the current stream and with each other"""
    if dev in _CAL:
        return _CAL[dev]
    if torch.cuda.is_current_stream_capturing():
        return []               # (no probing under stream capture; not remembered)
    picked = []
    if cfg.stream_probe and hasattr(torch.cuda, "_sleep"):
        main = torch.cuda.current_stream(dev)
        scratch = torch.zeros(64, device=f"cuda:{dev}")

        def overlap(a, b):
            torch.cuda.synchronize(dev)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(a):
                e0.record()
                torch.cuda._sleep(6000000)      # ~2.5 ms on the device
                e1.record()
            with torch.cuda.stream(b):
                scratch.add_(1.0)
                e2.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e2) < 0.5 * e0.elapsed_time(e1)
        try:
            torch.cuda._sleep(1000)
            for _ in range(10):
                c = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(c):      # first use of a stream creates its queue (milliseconds): not part of the probe
                    scratch.add_(1.0)
                if overlap(main, c) and all(overlap(x, c) for x in picked):
                    picked.append(c)
                    if len(picked) == 3:
                        break
        except Exception:       # the probe is an optimisation: any failure falls back to plain streams
            picked = []
    _CAL[dev] = picked
    return picked


_COMPUTE_STREAM = {}


def compute_stream(dev=None):
    """one non-default stream per device to compute on (torch.cuda.set_stream(ops.compute_stream())): on this runtime the process's
    default stream runs exclusively with respect to hipGraphs launched on other streams, so a step that replays its text towers
    from graphs must not compute on it (bench.py, train_stage1.main; data-parallel runs get theirs from place_streams)"""
    dev = torch.cuda.current_device() if dev is None else dev
    s = _COMPUTE_STREAM.get(dev)
    if s is None:
        s = _COMPUTE_STREAM[dev] = torch.cuda.Stream(device=dev)
    return s


def place_streams(group=None):
    """Data-parallel start-up (call once on every rank right after init_process_group, before anything else touches the
    device): sort out which stream runs on which of the four hardware queues WITH the collective backend in the picture.
    torch's NCCL/RCCL process group runs its kernels on a stream of its own whose queue nobody chooses -- if that is the
    compute stream's queue, every gradient all-reduce sits in front of the compute kernels until the peers arrive (measured
    with one rank: a collective issued while the compute stream sleeps completes only after the sleep), and the reducer's
    overlap is gone without any error.  Probes (device sleep + tiny kernel / tiny all-reduce, HIP-event timestamps) find the
    queue classes of a few candidate streams and the class of the backend's stream; compute, text and weight-gradient streams
    are then taken from three classes the backend is NOT on.  Every rank issues the same fixed number of probe collectives.
    Returns the stream the caller should make current (torch.cuda.set_stream) for all further work, or None to stay."""
    import torch.distributed as dist
    dev = torch.cuda.current_device()
    if not cfg.stream_probe or not hasattr(torch.cuda, "_sleep"):
        return None
    null = torch.cuda.current_stream(dev)
    scratch = torch.zeros(64, device=f"cuda:{dev}")
    cands = [null]
    torch.cuda._sleep(1000)
    for _ in range(7):
        c = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(c):
            scratch.add_(1.0)
        cands.append(c)
    torch.cuda.synchronize(dev)

    def probe(a, b, op):
        """does `op`, issued on b, finish while a is still asleep?"""
        torch.cuda.synchronize(dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(a):
            e0.record()
            torch.cuda._sleep(6000000)
            e1.record()
        with torch.cuda.stream(b):
            op()
            e2.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e2) < 0.5 * e0.elapsed_time(e1)
    reps, cls = [], []                      # one representative stream per hardware-queue class
    for c in cands:
        k = next((i for i, r in enumerate(reps) if r is c or not probe(r, c, lambda: scratch.add_(1.0))), None)
        if k is None:
            reps.append(c)
            k = len(reps) - 1
        cls.append(k)
    backend_cls = None
    if dist.is_initialized() and dist.get_backend(group) == "nccl":   # (no condition on what THIS rank found: same collectives everywhere)
        buf = torch.zeros(256, device=f"cuda:{dev}")
        for _ in range(2):                  # communicator / stream set-up is not part of the probe
            dist.all_reduce(buf, group=group)
        torch.cuda.synchronize(dev)
        for k in range(4):                  # a FIXED number of collectives on every rank, whatever the probes find locally
            r = reps[k % len(reps)]
            issue = reps[(k + 1) % len(reps)]

            def coll():
                w = dist.all_reduce(buf, group=group, async_op=True)
                w.wait()
            dist.barrier(group=group)       # ranks start each probe together: a late peer must not look like a shared queue
            if not probe(r, issue, coll) and backend_cls is None and k < len(reps):
                backend_cls = k
    if cfg.stream_probe_log:
        print(f"[place_streams] queue classes of the candidates {cls}, collective backend on class {backend_cls}", flush=True)
    free = [k for k in range(len(reps)) if k != backend_cls]
    if len(free) < 3:                       # fewer queues than roles: leave everything as the runtime handed it out
        return None
    order = ([cls[0]] if cls[0] in free else []) + [k for k in free if k != cls[0]]   # keep the current stream if it may stay
    compute, text, wgrad = reps[order[0]], reps[order[1]], reps[order[2]]
    # the reducer's ISSUE stream carries only event records / waits (the backend's own stream runs the kernels), but a wait
    # parked in a hardware queue blocks whatever is behind it there: it goes on the backend's queue, never the compute one's
    used = {cls[cands.index(compute)], cls[cands.index(text)], cls[cands.index(wgrad)]}
    issue = next((c for c, k in zip(cands, cls) if k == backend_cls), None)
    if issue is None:
        issue = next((c for c, k in zip(cands, cls) if k not in used), wgrad)
    _CAL[dev] = [text, wgrad, issue]
    _WG.pop(dev, None)
    for key in [k for k in _SIDE_STREAMS if k[0] == dev]:
        _SIDE_STREAMS.pop(key)
    return None if compute is null else compute


def _new_side_stream(dev, slot):
    """slot 0 = text towers, 1 = weight gradients, 2 = gradient reducer: distinct probed streams where the hardware has them"""
    pool = _calibrated_streams(dev)
    return pool[slot] if slot < len(pool) else torch.cuda.Stream(device=dev)


def _wgrad_stream():
    dev = torch.cuda.current_device()
    if dev not in _WG:
        _WG[dev] = _new_side_stream(dev, 1)
    return _WG[dev]


def on_wgrad_stream(fn, *tensors, sink=None):
    """Run `fn()` (kernel launches only) on the weight-gradient stream after everything queued so far on the current
    stream; `tensors` are the buffers it reads (kept alive for the side stream via record_stream); sink: the gradient-arena
    view it writes."""
    if _SEG is not None:
        return _SEG.defer(fn, tensors, sink)
    if not _wgrad_enabled():
        return fn()
    main, side = torch.cuda.current_stream(), _wgrad_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r = fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    return r


_SIDE_STREAMS = {}


def side_stream(name):
    """Named auxiliary HIP stream of the current device (e.g. "text": the text encoders overlap the RN50 trunk)."""
    key = (torch.cuda.current_device(), name)
    if key not in _SIDE_STREAMS:
        slot = {"text": 0, "reduce": 2}.get(name)
        _SIDE_STREAMS[key] = _new_side_stream(key[0], slot) if slot is not None else torch.cuda.Stream(device=key[0])
    return _SIDE_STREAMS[key]


def wgrad_join(into=None):
    """Make a stream (default: the current one) wait for every auxiliary stream that may still be producing gradients (the
    weight-gradient stream and the named side streams): call before anything consumes the gradient arenas (optimiser,
    all-reduce).  `into`: the stream that should wait -- the data-parallel reducer passes its own stream so that the COMPUTE
    stream is not stalled in the middle of backward."""
    if _SEG is not None:
        return  # segmented capture: the streams are joined between the graphs, by the object that replays them
    if not _WG and not _SIDE_STREAMS:
        return  # nothing was ever issued on an auxiliary stream (also: host-only / CPU test contexts)
    cur = torch.cuda.current_stream() if into is None else into
    dev = torch.cuda.current_device()
    if dev in _WG and _WG[dev] != cur:
        cur.wait_stream(_WG[dev])
    for (d, _), st in _SIDE_STREAMS.items():
        if d == dev and st != cur:
            cur.wait_stream(st)


def _sink(p):
    """The arena view a weight gradient should be written into, or None (=> return it to autograd)."""
    if p is not None and getattr(p, "_tris_sink", False):
        return p.grad
    return None


def _emit(p, make, needs, reads=None):
    """Produce a parameter gradient: into the sink (return None) or as a fresh tensor.  reads: the tensors `make` reads -- a gradient
    that lands in the arena and that nothing downstream of the backward pass waits for (a bias or norm-parameter column sum) is then
    issued on the weight-gradient stream like the weight gradients themselves, off the compute stream's chain (cfg.side_param_grads)."""
    if not needs:
        return None
    s = _sink(p)
    if s is not None:
        if reads is not None and cfg.side_param_grads:
            on_wgrad_stream(lambda: make(s), *reads, sink=s)
        else:
            make(s)
        return None
    out = torch.empty_like(p, memory_format=torch.preserve_format)
    make(out)
    return out


# ----------------------------------------------------------------------------------------------- profiling hook
# bench.py measures the dominant kernel family live: when enabled, every MFMA GEMM / implicit-GEMM conv launch is
# bracketed by HIP events on the launch stream and logged with its algorithmic FLOPs.
_PROF = None
_PROF_SHAPES = False


def profile_begin():
    global _PROF
    _PROF = []


def profile_end():
    """-> list of (kind, flops, milliseconds, algorithmic bytes)"""
    global _PROF
    rec, _PROF = _PROF, None
    torch.cuda.synchronize()
    return [(k, f, a.elapsed_time(b), nb) for k, f, a, b, nb in rec]


def _shape_tag(text):
    return text if _PROF_SHAPES else None


def _timed(kind, flops, fn, tag=None, nbytes=0):
    """nbytes: ALGORITHMIC HBM bytes of the launch -- every operand read once, every output written once"""
    if _PROF is None:
        return fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = fn()
    b.record()
    if r is not False:   # (False = the entry point declined the shape: nothing was launched)
        _PROF.append((kind if tag is None else f"{kind}:{tag}", flops, a, b, nbytes))
    return r


# ----------------------------------------------------------------------------------------------- raw launches
# Batch-invariant arithmetic (evaluation): with it on, the result of every output ROW of a dense product is independent of how
# many other rows share the launch.  What can differ between launches of different M is (a) split-K (a different partition of
# the K sum) and (b) the direct vs implicit 3x3 convolution (different K order); tile shapes and the classic / pipelined loops
# walk K in the same order and are bitwise neutral.  `batch_invariant()` switches (a) and (b) off for its scope, so that
# tris_amd.validate can batch images and sentences and still return exactly the numbers of the one-at-a-time loop.
_BATCH_INVARIANT = 0


class batch_invariant:
    def __enter__(self):
        global _BATCH_INVARIANT
        _BATCH_INVARIANT += 1
        if _BATCH_INVARIANT == 1:
            call("tris_set_conv_direct_thread", 0)     # this thread's products only; nothing process-wide is touched
        # h2 scales every operand by a power of two derived from the amax of the WHOLE tensor: a row's rounding depends on which other
        # rows share the launch, so h2 cannot be batch-invariant (measured in round 6: one of 128 hit-tests flipped between group
        # sizes).  Inside this scope products run in the split-bf16 x3 arithmetic -- exact pieces, no data-dependent scale, row results
        # independent of the batch -- which is also the faster one at evaluation sizes (no amax words to produce or read).
        _H2["paused"] += 1
        return self

    def __exit__(self, *exc):
        global _BATCH_INVARIANT
        _BATCH_INVARIANT -= 1
        _H2["paused"] -= 1
        if _BATCH_INVARIANT == 0:
            call("tris_set_conv_direct_thread", -1)
        return False


def gemm(A, B, C, M, N, K, lda, ldb, ldc, tA, tB, batch=1, sA=0, sB=0, sC=0, bias=None, bias_mode=0, resid=None,
         ldr=0, sR=0, act=0, alpha=1.0, use_ws=True, pre_out=None, dact=None, w_a=False, w_b=False, w_bt=False, mark=True):
    """pre_out / dact (tris_gemm_epilogue_next): also store the pre-activation value / multiply the result by quickgelu'(dact);
    the call then returns False -- nothing launched -- when the fast kernel does not serve the operands (caller falls back).
    w_a / w_b: that operand is a convolution weight, which exists as operand planes when the other operand is a plane tensor.
    mark=False: no product will read C as an operand (a weight gradient written into the optimiser's arena): the launch -- and the
    split-K reduce behind it -- does not carry the amax by-product (the reduce was 65 % longer with it)."""
    _chk(A, B, C, bias, resid)
    if pre_out is not None or dact is not None or batch != 1 or _BATCH_INVARIANT:
        A, B = unplanes(A), unplanes(B)
    if pre_out is not None or dact is not None:
        if _BATCH_INVARIANT or batch != 1:
            return False
        h2_arm(A, B)
        h2_mark_next(C)
        call("tris_gemm_epilogue_next", P(pre_out), P(dact))
        return _timed("gemm", 2.0 * M * N * K, lambda: _declinable(
            "tris_gemm_f32", P(A), P(B), P(C), M, N, K, lda, ldb, ldc, int(tA), int(tB), 1, 0, 0, 0, P(bias), bias_mode, P(resid),
            ldr, sR, act, float(alpha), None, 0, _stream()),
            nbytes=4.0 * (M * K + K * N + M * N * (2 + int(resid is not None))))
    if _BATCH_INVARIANT and batch == 1 and not tA and M < 4 and lda == K and ldc == N and (resid is None or ldr == N) and bias_mode != 2:
        # products of fewer than four rows run a different (generic) kernel than larger ones: in batch-invariant mode they are
        # padded to four rows so that a sentence evaluated alone goes through the same arithmetic as one evaluated in a batch
        A4 = A.new_zeros(4, K)
        A4[:M].copy_(A.reshape(M, K))
        R4 = None
        if resid is not None:
            R4 = resid.new_zeros(4, N)
            R4[:M].copy_(resid.reshape(M, N))
        C4 = C.new_empty(4, N)
        gemm(A4, B, C4, 4, N, K, K, ldb, N, False, tB, bias=bias, bias_mode=bias_mode, resid=R4, ldr=N, act=act, alpha=alpha,
             use_ws=use_ws)
        C.reshape(M, N).copy_(C4[:M])
        return C
    ws = workspace(0) if (use_ws and batch == 1 and not _BATCH_INVARIANT) else None   # (no workspace = no split-K)
    if (w_b and batch == 1 and not tA and M >= cfg.gemm_convert and cfg.gemm_convert > 0 and K % 32 == 0 and lda == K and getattr(B, "_tris_linear_w", False)
            and planes_on() and pl_word(A) is None and A.is_contiguous() and A.data_ptr() % 16 == 0 and _pl_weight_ok(B)):
        A = _a_planes(A, M, K)
    pA, pB = h2_pp(A, B, w_a=w_a, w_b=w_b, k_red=K, w_bt=w_bt and not tB) if batch == 1 else (P(A), P(B))
    if batch == 1 and PL_STATS["last_t"]:    # B = W [K][N] arrived as the planes of its transpose [N][K]
        tB, ldb = True, K
    if mark:
        h2_mark_next(C)  # (h2: the product also leaves the amax of what it writes -- another product may consume C directly)
    if ws is not None and cfg.fuse_splitk:
        tk = splitk_tickets()
        if tk is not None:
            call("tris_splitk_tickets_next", tk.data_ptr(), tk.numel())
    _timed("gemm", 2.0 * M * N * K * batch, lambda: call(
        "tris_gemm_f32", pA, pB, P(C), M, N, K, lda, ldb, ldc, int(tA), int(tB), batch, sA, sB, sC, P(bias),
        bias_mode, P(resid), ldr, sR, act, float(alpha), P(ws), 0 if ws is None else ws.numel() * 4, _stream()),
        tag=(f"{'T' if tA else 'N'}{'T' if tB else 'N'} M{M} N{N} K{K} b{batch}" if _PROF_SHAPES else None),
        nbytes=4.0 * ((M * K if (sA == 0 and batch > 1) else batch * M * K) + batch * K * N
                      + batch * M * N * (2 if resid is not None else 1)))
    return C


def colsum(X, M, N, out):
    ws = workspace(query("tris_col_workspace_bytes", M, N))
    call("tris_colsum_f32", P(X), M, N, N, P(out), P(ws), _stream())
    return out


def ew(op, A, B=None, s=0.0, out=None):
    _chk(A, B)
    out = torch.empty_like(A) if out is None else out
    h2_mark_next(out)
    call("tris_elementwise_f32", EW[op], P(A), P(B), P(out), A.numel(), float(s), _stream())
    return out


def nchw_to_nhwc(x):
    _chk(x)
    if x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous():   # already channels-last in memory (tris_amd.dataset.hbm)
        return x.permute(0, 2, 3, 1)
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty(B, H, W, C, device=x.device, dtype=x.dtype)
    call("tris_nchw_to_nhwc_f32", P(x), P(y), B, C, H, W, _stream())
    return y


# ----------------------------------------------------------------------------------------------- fused BN statistics
# A conv forward can emit the BatchNorm batch statistics of its output from the GEMM epilogue (tris_*_bnstat_f32).
# The fp64 partials travel from the conv Function to the BatchNorm Function as an attribute of the output tensor
# (`y._bn_part`, set inside the Function's forward: the object returned by apply() is that same tensor).
def _launch_with_stats(y, M, N, launch):
    import ctypes
    part = torch.empty(((M + 127) // 128) * 2 * N, device=y.device, dtype=torch.float64)
    rows = ctypes.c_int(0)
    launch(part, ctypes.byref(rows))
    if rows.value > 0:
        y._bn_part = (part, rows.value)


def _declinable(name, *args):
    """call an entry point that may decline a shape: True if it ran, False if it returned TRIS_DECLINED (caller falls back)"""
    err = getattr(_lib.load(), name)(*args)
    if err == -2:   # TRIS_DECLINED
        return False
    if err != 0:
        raise _lib.TrisHipError(f"{name} failed with hipError_t {err}")
    return True


class GradBox:
    """Hand-off of a residual-branch gradient inside one block.  In `out = relu(bn3(f(x)) + x)` the tensor x has two
    consumers (f's first conv and the residual input of bn3); autograd would sum their two gradients with an extra
    elementwise pass over the activation.  Instead bn3's backward -- which always runs before the first conv's, because
    the latter depends on it through f -- leaves its residual gradient here, and the first conv's data-gradient GEMM adds
    it in its epilogue (`resid`).  Used by the identity Bottlenecks of the RN50 trunk.

    In the down-sampling Bottlenecks the second consumer of x is the shortcut (avg-pool / 1x1 conv); its backward normally
    runs before conv1's too (autograd schedules later-created nodes first) but nothing guarantees it, so the depositor
    checks `consumed` and simply returns its gradient to autograd if conv1 has already run."""
    __slots__ = ("value", "consumed")

    def __init__(self):
        self.value = None
        self.consumed = False

    def deposit(self, g):
        """-> True if the gradient was taken over (the caller must then return None for that input)"""
        if self.consumed or self.value is not None:
            return False
        self.value = g
        return True


# ----------------------------------------------------------------------------------------------- Linear / 1x1 conv
# ---- "h2" products (ops.set_gemm_mode("h2") / TRIS_GEMM_MODE=h2) -------------------------------------------
# Every product of the GEMM / convolution family with two fp16 pieces per operand and three f16 MFMAs (half the matrix work of
# x3).  fp16 holds 11 significand bits per piece but only 5 exponent bits, so every operand is scaled by a power of two derived --
# inside the kernel -- from the largest magnitude of its tensor; the residual piece is stored pre-scaled by 2^11, which keeps the
# representation fp32-class over a 2^27 range inside a tensor (csrc/x3_split.h).  The amax is a word in a pool, written
#   * by the pass that PRODUCES the tensor (BatchNorm apply / backward apply, LayerNorm, elementwise: h2_mark_next, no extra traffic),
#   * for all parameters of an optimiser arena by ONE launch per step; for other parameters (a frozen model) once, into a pool of
#     constants, until the parameter is written to,
#   * for a BatchNorm + ReLU folded into its consuming convolution -- a tensor that never exists -- as an upper bound from the affine
#     parameters (h2_bound_next),
#   * otherwise by a small pre-pass (tensors up to 128 MB).
# A product whose operands have no amax runs in x3.  A "step" is the life of the pool's words: h2_begin_step() clears the pool
# (train_step calls it; any other entry -- an evaluation forward, a lone op in a test -- starts one on first use or when the pool
# is full).
_H2 = {"pool": None, "next": 0, "step": 0, "paused": 0, "arenas": [], "const": None, "const_next": 0, "const_tags": {}}
_H2_STEP_IDS = [0]       # step ids are unique across the main pool and private pools: a tag never matches another pool's step
H2_SLOTS = 4096          # amax words per step; a word is H2_SUB unsigned words (csrc/amax.h amax_raise)
H2_SUB = 2048
H2_CONST_SLOTS = 1024
H2_PREPASS_MAX = 1 << 25
H2_AUTO_KEEP = 4         # retired pools of gradient-mode forwards issued outside an explicit step (h2_begin_step)


def h2_on():
    return _ARITH == "h2" and not _H2["paused"] and not _thread_mode_is_set()


_H2_ARENAS = []   # weak references to the optimiser arenas whose parameters get their amax from ONE launch per step


def h2_register_arena(arena):
    """an optimiser arena (tris_amd.optim.Arena: .p flat parameter buffer, .params, .offsets).  Its parameters are updated in
    place by kernels that do not bump torch's version counters, so they never get a CONSTANT amax word (_h2_const_slot)."""
    import weakref
    r = weakref.ref(arena)
    _H2_ARENAS.append(r)
    for p in arena.params:
        p._tris_arena = r       # (the LATEST arena a parameter was put into: an optimiser built later supersedes an earlier one)


def _h2_live_arenas():
    """the registered arenas that still own their parameters (an optimiser object that was superseded by a newer one over the same
    parameters, or is only kept alive by a reference cycle, is not one), newest first, as many as fit half the pool"""
    out, words = [], 0
    for r in reversed(list(_H2_ARENAS)):
        a = r()
        if a is None or not a.params or getattr(a.params[0], "_tris_arena", None) is not r:
            _H2_ARENAS.remove(r)
        elif _H2["pool"] is not None and a.p.device == _H2["pool"].device and words + len(a.params) <= H2_SLOTS // 2:
            out.append(a)
            words += len(a.params)
    return out


def h2_weights_amax(arenas=None):
    """(re)compute the amax words of the arenas' parameters: the first words of the pool, in order.  arenas: the list a
    captured step recorded (its launches read those words); default: the live registered ones"""
    base = 0
    for a in (_h2_live_arenas() if arenas is None else arenas):
        if not hasattr(a, "_h2_dev"):
            a._h2_dev = (torch.tensor(list(a.offsets), device=a.p.device, dtype=torch.int64),
                         torch.tensor([p.numel() for p in a.params], device=a.p.device, dtype=torch.int64))
        call("tris_amax_segments_f32", P(a.p), a._h2_dev[0].data_ptr(), a._h2_dev[1].data_ptr(), len(a.params),
             _H2["pool"].data_ptr() + 4 * H2_SUB * base, _stream())
        if cfg.h2_planes:
            # operand planes of the convolution weights (csrc/planes.h), refreshed with the amaxes: one more launch per arena
            if not hasattr(a, "_pl_dev"):
                idx = [i for i, p in enumerate(a.params) if _pl_weight_ok(p)]
                a._pl_idx = idx
                a._pl_dev = None if not idx else (
                    torch.tensor([a.offsets[i] for i in idx], device=a.p.device, dtype=torch.int64),
                    torch.tensor([a.params[i].numel() for i in idx], device=a.p.device, dtype=torch.int64),
                    torch.tensor(idx, device=a.p.device, dtype=torch.int64))
                a.pl = torch.zeros_like(a.p) if idx else None
            if a._pl_dev is not None:
                call("tris_h2_planes_segments_f32", P(a.p), a._pl_dev[0].data_ptr(), a._pl_dev[1].data_ptr(), a._pl_dev[2].data_ptr(),
                     len(a._pl_idx), _H2["pool"].data_ptr() + 4 * H2_SUB * base, P(a.pl), _stream())
            # ... and of the 1x1 weights TRANSPOSED: their data gradient dX = dY . W is then the forward's row-major A x B^T product
            if not hasattr(a, "_plt_dev"):
                idt = [i for i in getattr(a, "_pl_idx", []) if _pl_weight_t_ok(a.params[i])]
                a._plt_idx = idt
                mk = lambda v: torch.tensor(v, device=a.p.device, dtype=torch.int64)
                a._plt_dev = None if not idt else (mk([a.offsets[i] for i in idt]), mk([a.params[i].shape[0] for i in idt]),
                                                   mk([a.params[i].shape[1] for i in idt]), mk(idt))
                a.plt = torch.zeros_like(a.p) if idt else None
            if a._plt_dev is not None:
                call("tris_h2_planes_t_segments_f32", P(a.p), a._plt_dev[0].data_ptr(), a._plt_dev[1].data_ptr(), a._plt_dev[2].data_ptr(),
                     a._plt_dev[3].data_ptr(), len(a._plt_idx), _H2["pool"].data_ptr() + 4 * H2_SUB * base, P(a.plt), _stream())
        base += len(a.params)
    return base


def _pl_weight_t_ok(p):
    """a 1x1-convolution weight [Cout, Cin, 1, 1] whose transposed planes exist (64 x 64 tiles of the transposing pass)"""
    if p.dim() == 2:     # a Linear weight [N, K] of a tower whose products run on planes (VisionTransformer marks them)
        return cfg.gemm_convert > 0 and getattr(p, "_tris_linear_w", False) and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0
    return p.dim() == 4 and p.shape[2] * p.shape[3] == 1 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0


def _pl_weight_ok(p):
    """a parameter that dense products of the trunk read as fp16 operand planes: a convolution weight [Cout, Cin, kh, kw] whose
    contiguous dimension (Cin: channels_last memory) is a multiple of 8"""
    if p.dim() == 2:     # (round 6) a Linear weight [N, K], K contiguous: the same geometry as a 1x1 convolution's
        return cfg.gemm_convert > 0 and getattr(p, "_tris_linear_w", False) and p.shape[1] % 8 == 0 and p.is_contiguous()
    return p.dim() == 4 and p.shape[1] % 8 == 0 and (p.is_contiguous(memory_format=torch.channels_last) or p.shape[2] * p.shape[3] == 1)


def _a_planes(A, M, K):
    """(round 6, cfg.gemm_convert) The row-major A operand [M, K] of a large product against a weight that exists as planes, converted
    to planes ONCE in a pass of its own (tris_h2_planes_f32: 4 bytes read + 4 written per element) so that the product runs on the
    LDS-DMA loop with no split in its K loop -- every column tile of the product would otherwise re-split the same rows (a ViT-B/16
    trunk product has 6 ... 24 of them).  Measured per shape in profiles/r6_vit_planes_bench.txt (x1.2 ... 1.57 for the product, 5 ... 13 us
    for the pass at 2400 rows).  Returns the plane tensor (tagged with A's amax word) or A itself where A has no word."""
    slot = _h2_amax(A)
    if slot is None:
        return A
    t = torch.empty_like(A)
    call("tris_h2_planes_f32", P(A), P(t), A.numel(), slot, _stream())
    PL_STATS["converted"] = PL_STATS.get("converted", 0) + 1
    return pl_tag(t, slot)


def _h2_new_step_id():
    _H2_STEP_IDS[0] += 1
    return _H2_STEP_IDS[0]


def h2_begin_step(explicit=True):
    """A fresh amax pool (one memset) + the weights' amaxes (one launch per optimiser arena).  Returns the pool (None when the
    arithmetic is not h2).  Whatever was issued earlier on a side stream may still read the words about to be cleared: the
    current stream first waits for the side streams.  explicit: the caller brackets a whole step (h2_end_step closes it)."""
    if _ARITH != "h2":
        return None
    _H2["in_step"] = bool(explicit)
    wgrad_join()
    dev = torch.device("cuda", torch.cuda.current_device())
    # A forward issued OUTSIDE an explicit step with gradients on may still have its autograd graph -- and the plane tensors saved in
    # it -- alive when the next such forward begins (gradient accumulation, a drop-in user summing the losses of two forwards before
    # one backward: legal PyTorch).  Its pool is then RETIRED, not cleared: the words stay what the saved plane tensors were written
    # under (pl_word accepts the H2_AUTO_KEEP most recent retired steps) and the new forward gets a fresh pool.
    live = _H2.setdefault("auto_pools", {})
    if _H2["pool"] is None or _H2["pool"].device != dev:
        _H2["pool"] = torch.zeros(H2_SLOTS * H2_SUB, device=dev, dtype=torch.int32)
        live.clear()
    elif not explicit and _H2.get("auto_grad") and not _H2.get("private") and not torch.cuda.is_current_stream_capturing():
        live[_H2["step"]] = _H2["pool"]
        while len(live) > H2_AUTO_KEEP:
            live.pop(next(iter(live)))
        _H2["pool"] = torch.zeros(H2_SLOTS * H2_SUB, device=dev, dtype=torch.int32)
    else:
        if explicit:
            live.clear()
        _H2["pool"].zero_()
    _H2["auto_grad"] = (not explicit) and torch.is_grad_enabled() and not _H2.get("private")
    _H2["step"] = _h2_new_step_id()
    _H2["plane_numel"] = {}
    _PL_GRAD.clear()
    base = 0
    arenas = _h2_live_arenas() if not _H2.get("private") else []
    for a in arenas:
        for i, p in enumerate(a.params):
            p._h2 = (_H2["step"], _H2["pool"].data_ptr() + 4 * H2_SUB * (base + i), p._version)
        base += len(a.params)
    _H2["next"] = _H2["base"] = base
    _H2["arenas"] = arenas
    h2_weights_amax(arenas)
    if cfg.h2_planes:
        for a in arenas:
            for i in (a._pl_idx if getattr(a, "pl", None) is not None else ()):
                p = a.params[i]
                p._plw = (_H2["step"], a.pl.data_ptr() + 4 * a.offsets[i], p._h2[1])
                _H2["plane_numel"][p._h2[1]] = p.numel()
            for i in (a._plt_idx if getattr(a, "plt", None) is not None else ()):
                p = a.params[i]
                p._plwt = (_H2["step"], a.plt.data_ptr() + 4 * a.offsets[i], p._h2[1])
    return _H2["pool"]


def h2_end_step():
    """the explicit step ends: the words stay valid until the next begin; forwards issued outside a step start their own again"""
    _H2["in_step"] = False


def h2_auto_step():
    """Called where a training forward STARTS (TRIS.forward, ModifiedResNet.forward_cl).  With operand planes a forward must not
    straddle two pools -- a plane tensor is only meaningful with the word of the pool it was written under -- so a forward issued
    OUTSIDE an explicit step (train_step brackets its own with h2_begin_step / h2_end_step) begins a fresh step of its own instead
    of running into a half-used pool that restarts mid-way.  No-op inside an explicit step, when planes are off, and when nothing
    has been handed out since the last begin.  Returns True if a step was begun (the caller's nested forwards must not begin another:
    h2_auto_lock)."""
    if not planes_on() or _H2.get("in_step") or _H2.get("auto_lock") or _H2.get("private"):
        return False
    if _H2["pool"] is not None and _H2["next"] == _H2.get("base", -1):
        return False
    h2_begin_step(explicit=False)
    return True


class h2_auto_lock:
    """inside: nested forwards do not begin a step of their own (TRIS.forward around its trunk call)"""

    def __enter__(self):
        self.prev = _H2.get("auto_lock", False)
        _H2["auto_lock"] = True

    def __exit__(self, *a):
        _H2["auto_lock"] = self.prev


class h2_private_pool:
    """Products launched inside take their per-step amax words from a pool of their own, cleared on entry: a region that is
    captured into a hipGraph of its own and replayed outside the steps it was recorded in (the frozen aux text tower) must not
    hold words of the per-step pool.  Parameters inside get constant words (they must not be arena parameters of a live step)."""

    def __init__(self, slots=512):
        self.slots, self.pool = slots, None

    def __enter__(self):
        self.active = _ARITH == "h2"
        if not self.active:
            return self
        self.saved = {k: _H2.get(k) for k in ("pool", "next", "step", "arenas", "private", "limit")}
        dev = torch.device("cuda", torch.cuda.current_device())
        if self.pool is None or self.pool.device != dev:
            self.pool = torch.zeros(self.slots * H2_SUB, device=dev, dtype=torch.int32)
        _H2.update(pool=self.pool, arenas=[], private=True, limit=self.slots)
        self.reset()
        return self

    def reset(self):
        """clear the words and hand them out from the first again (call it as the first thing INSIDE a capture of the region: the
        memset is then part of the graph, and the words are handed out in the same order at every recording)"""
        if self.active:
            self.pool.zero_()
            _H2.update(next=0, step=_h2_new_step_id())

    def __exit__(self, *exc):
        if self.active:
            _H2.update(self.saved)
        return False


class h2_paused:
    """products launched inside run in x3"""

    def __enter__(self):
        _H2["paused"] += 1

    def __exit__(self, *a):
        _H2["paused"] -= 1


def _h2_slot():
    limit = _H2.get("limit") or H2_SLOTS
    if _H2["pool"] is None or (_H2["next"] >= limit and not _H2.get("private")):
        h2_begin_step(explicit=bool(_H2.get("in_step")))          # first use outside a training step, or the pool is full: a new step
    i = _H2["next"]
    if i >= limit:
        return None              # (a private pool is never restarted: its region would lose words it still reads)
    _H2["next"] = i + 1
    return _H2["pool"].data_ptr() + 4 * H2_SUB * i


def h2_mark_next(t):
    """call right BEFORE the launch that writes `t` (BatchNorm apply-type passes, LayerNorm, elementwise): that launch also
    leaves t's amax in a pool word"""
    if t is None or not h2_on():
        return
    slot = _h2_slot()
    if slot is None:
        return
    call("tris_amax_next", slot)
    t._h2 = (_H2["step"], slot, t._version)


def h2_bound_word(gamma, beta, C, xhat_max):
    """amax word holding an upper bound of |relu(bn(x))| for a train-mode BatchNorm whose output is never written (None: not h2)"""
    if not h2_on():
        return None
    slot = _h2_slot()
    if slot is not None:
        call("tris_bn_out_bound_f32", P(gamma), P(beta), C, float(xhat_max), slot, _stream())
    return slot


def _h2_tag_slot(t):
    tag = getattr(t, "_h2", None)
    if tag is not None and tag[0] == _H2["step"] and tag[2] == t._version:
        return tag[1]
    return None


def _h2_const_slot(t):
    """amax word of a parameter outside the optimiser arenas (a frozen model's): computed once, valid until it is written to"""
    key = id(t)
    tag = _H2["const_tags"].get(key)
    if tag is not None and tag[1] == t.data_ptr() and tag[3]() is t and _H2["const"] is not None and _H2["const"].device == t.device:
        if tag[0] == t._version:
            return tag[2]
        # written to since (load_state_dict, seed_fill): refresh the SAME word on the stream -- a hipGraph that recorded its address
        # (the frozen aux text tower, the evaluation graphs) goes on reading a valid amax, and no new slot is used up
        i = (tag[2] - _H2["const"].data_ptr()) // (4 * H2_SUB)
        _H2["const"][i * H2_SUB:(i + 1) * H2_SUB].zero_()
        call("tris_amax_bits_f32", P(t), t.numel(), tag[2], _stream())
        _H2["const_tags"][key] = (t._version, t.data_ptr(), tag[2], tag[3])
        return tag[2]
    if _H2["const"] is None or _H2["const"].device != t.device:
        _H2["const"] = torch.zeros(H2_CONST_SLOTS * H2_SUB, device=t.device, dtype=torch.int32)
        _H2["const_next"] = 0
        _H2["const_tags"].clear()
    if _H2["const_next"] >= H2_CONST_SLOTS:
        return None
    import weakref
    slot = _H2["const"].data_ptr() + 4 * H2_SUB * _H2["const_next"]
    _H2["const_next"] += 1
    call("tris_amax_bits_f32", P(t), t.numel(), slot, _stream())
    _H2["const_tags"][key] = (t._version, t.data_ptr(), slot, weakref.ref(t))
    return slot


def _h2_amax(t):
    slot = _h2_tag_slot(t)
    if slot is not None:
        return slot
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    if not (t.numel() <= H2_PREPASS_MAX and dense and t.data_ptr() % 16 == 0 and t.numel() > 0):
        return None
    if isinstance(t, torch.nn.Parameter) and not getattr(t, "_tris_arena", False):
        slot = _h2_const_slot(t)
        if slot is not None:
            return slot
    slot = _h2_slot()
    if slot is None:
        return None
    call("tris_amax_bits_f32", P(t), t.numel(), slot, _stream())
    try:
        t._h2 = (_H2["step"], slot, t._version)
    except AttributeError:
        pass
    return slot


def h2_arm(A, B, a_slot=None, b_slot=None):
    """before a dense product of A and B: run it in h2 if both operands have an amax (one shot, this thread).  a_slot / b_slot: the
    operand's amax word where the tensor itself is not at hand (a BatchNorm folded into the consuming convolution)"""
    if not h2_on():
        return False
    a = a_slot if a_slot is not None else (_h2_amax(A) if A is not None else None)
    b = None
    if a is not None:
        b = b_slot if b_slot is not None else (_h2_amax(B) if B is not None else None)
    if a is None or b is None:
        return False
    call("tris_h2_next", a, b, 0.0, 0.0)
    return True


# ---- operand planes (csrc/planes.h, DESIGN.md "operand planes") --------------------------------------------------------------
# In h2 the activations of the RN50 trunk between a BatchNorm and the convolutions that read it, the gradients between a BatchNorm
# backward and the convolution backward that reads it, and the convolution weights exist as fp16 PIECE PLANES: float32 tensors in
# torch's eyes whose bytes are the two fp16 pieces of every element, scaled by the power of two their amax word implies.  The tensor
# carries `_pl = (step, word)`; only the ops below that say so accept one, everything else must see `unplanes(t)`.
PL_STATS = {"unplanes": 0, "dx_planes": 0, "dy_planes": 0, "products": 0, "mixed": 0, "last": False, "last_t": False}


_HAS_GPU = None


def planes_on():
    global _HAS_GPU
    if _HAS_GPU is None:
        _HAS_GPU = torch.cuda.is_available()     # (host-logic tests run the model's control flow on CPU stand-ins)
    return _HAS_GPU and cfg.h2_planes and h2_on() and not _BATCH_INVARIANT


def pl_word(t):
    """amax word of a plane tensor (None: `t` is an ordinary fp32 tensor)"""
    tag = getattr(t, "_pl", None) if t is not None else None
    if tag is None:
        return None
    if (tag[0] != _H2["step"] and tag[0] not in _H2.get("auto_pools", ())) or tag[2] != t.data_ptr():
        raise RuntimeError("a plane tensor outlived the step whose amax pool scales it")
    return tag[1]


def pl_tag(t, word):
    t._pl = (_H2["step"], word, t.data_ptr())
    t._h2 = (_H2["step"], word, t._version)
    _H2.setdefault("plane_numel", {})[word] = t.numel()
    return t


def h2_range_report(threshold=0.01):
    """Range tell-tale of the plane tensors WRITTEN since the last h2_begin_step (host sync; csrc/planes.h): an element is held with
    its full 22 bits only down to 2^-27 of its tensor's scaled bound; the writing passes count the non-zero elements below that floor.
    -> {"plane_tensors", "elements", "elements_below_floor", "max_fraction_below_floor", "out_of_range_operands"} where the last is the
    number of tensors with more than `threshold` of their elements in the absolute-accuracy regime (bench.py prints it as
    h2_out_of_range_operands; the B = 48 parity test asserts 0)."""
    numel = _H2.get("plane_numel") or {}
    pool = _H2["pool"]
    if pool is None or not numel:
        return {"plane_tensors": 0, "elements": 0, "elements_below_floor": 0, "max_fraction_below_floor": 0.0, "out_of_range_operands": 0}
    torch.cuda.synchronize()
    base = pool.data_ptr()
    words = sorted(numel)
    idx = torch.tensor([(w - base) // (4 * H2_SUB) for w in words], device=pool.device, dtype=torch.int64)
    cnt = pool.view(-1, H2_SUB)[idx][:, 1::16].to(torch.int64).sum(1).cpu().tolist()
    fr = [c / max(numel[w], 1) for c, w in zip(cnt, words)]
    return {"plane_tensors": len(words), "elements": int(sum(numel.values())), "elements_below_floor": int(sum(cnt)),
            "max_fraction_below_floor": max(fr), "out_of_range_operands": sum(1 for f in fr if f > threshold)}


def _unplanes_raw(t, cache=True):
    w = pl_word(t)
    c = getattr(t, "_pl_f32", None) if cache else None
    if c is None:
        c = torch.empty_like(t)
        call("tris_h2_unplanes_f32", P(t), P(c), t.numel(), w, _stream())
        c._h2 = (_H2["step"], w, c._version)   # (the word bounds the rebuilt values as well)
        if cache:
            t._pl_f32 = c
        PL_STATS["unplanes"] += 1
    return c


class _UnplanesFn(torch.autograd.Function):
    """differentiable form of unplanes: the gradient of the rebuilt tensor IS the gradient of the plane tensor (fp32 either way)"""

    @staticmethod
    def forward(ctx, t):
        return _unplanes_raw(t, cache=False)

    @staticmethod
    def backward(ctx, g):
        return g


def unplanes(t):
    """the fp32 tensor a plane tensor stands for (its h2 operand rounding: 22 significand bits); fp32 tensors pass through"""
    if pl_word(t) is None:
        return t
    if t.requires_grad and torch.is_grad_enabled():
        return _UnplanesFn.apply(t)
    return _unplanes_raw(t)


_PL_CONST = {}   # id(parameter) -> (version, data_ptr, word, planes tensor, weakref): parameters outside the optimiser arenas


def _pl_weight(w):
    """(pointer to the planes of a convolution weight, its amax word) or None"""
    tag = getattr(w, "_plw", None)
    if tag is not None and tag[0] == _H2["step"]:
        return tag[1], tag[2]
    if not _pl_weight_ok(w):
        return None
    word = _h2_amax(w)
    if word is None:
        return None
    c = _PL_CONST.get(id(w))
    if c is not None and c[0] == w._version and c[1] == w.data_ptr() and c[2] == word and c[4]() is w:
        return c[3].data_ptr(), word
    import weakref
    wc = cl_weight(w)
    pl = torch.empty(wc.numel(), device=w.device, dtype=torch.float32)
    call("tris_h2_planes_f32", wc.data_ptr(), P(pl), wc.numel(), word, _stream())
    if len(_PL_CONST) > 4096:
        _PL_CONST.clear()
    _PL_CONST[id(w)] = (w._version, w.data_ptr(), word, pl, weakref.ref(w))
    return pl.data_ptr(), word


def _retag(t, tag):
    """a tensor that comes back from ctx.saved_tensors keeps its plane tag (set again in case autograd re-wrapped the object)"""
    if tag is not None and t is not None and getattr(t, "_pl", None) is None:
        t._pl = tag
        t._h2 = (tag[0], tag[1], t._version)
    return t


_PL_GRAD = {}   # data_ptr -> (step, word, shape): gradients a BatchNorm backward wrote as planes, until their one consumer takes them


_PL_GRAD_CB = [False]


def _pl_grad_end_of_backward():
    """ runs when the autograd pass that registered plane gradients ends: an entry that is still there found no
    plane-aware consumer -- its bytes were read as fp32 by whoever took the tensor, and the address-keyed tag would wait for an
    unrelated gradient that reuses the address.  Never silent: drop the entries and raise."""
    _PL_GRAD_CB[0] = False
    if _PL_GRAD:
        left = [(hex(k), v[2]) for k, v in _PL_GRAD.items()]
        _PL_GRAD.clear()
        raise RuntimeError(f"plane gradients without a plane-aware consumer at the end of backward: {left[:4]}")


def pl_grad_out(t, word):
    pl_tag(t, word)
    _PL_GRAD[t.data_ptr()] = (_H2["step"], word, tuple(t.shape))
    PL_STATS["dx_planes"] += 1
    if not _PL_GRAD_CB[0]:
        try:   # (inside an autograd pass: BatchNormFn.backward calls this)
            torch.autograd.Variable._execution_engine.queue_callback(_pl_grad_end_of_backward)
            _PL_GRAD_CB[0] = True
        except RuntimeError:
            pass
    return t


def pl_grad_in(dy):
    """the gradient a backward is handed: tag it as a plane tensor if a BatchNorm backward wrote it as one (one consumer: the entry
    is taken).  Returns dy."""
    ent = _PL_GRAD.pop(dy.data_ptr(), None)
    if ent is not None and ent[0] == _H2["step"] and ent[2] == tuple(dy.shape):
        if getattr(dy, "_pl", None) is None:
            dy._pl = (ent[0], ent[1], dy.data_ptr())
            dy._h2 = (ent[0], ent[1], dy._version)
        PL_STATS["dy_planes"] += 1
    elif getattr(dy, "_pl", None) is not None:
        raise RuntimeError("a plane gradient reached a second consumer")
    return dy


def h2_pp(A, B, a_slot=None, b_slot=None, y_planes=False, w_a=False, w_b=False, k_red=32, w_bt=False, y_mask=False):
    """Arm the next dense product for h2 (as h2_arm) and return the pointers of its operands A, B to hand to the entry point.  With
    operand planes on: if one operand IS a plane tensor and the other one is too, or is a convolution weight (w_a / w_b: that
    operand is a parameter), the product runs on planes; a plane tensor next to an fp32 operand is rebuilt (counted in PL_STATS).
    k_red: the reduction length of the product -- the plane kernels (the fast kernels) need a multiple of 32; anything else (the weight
    gradients of a 10 x 10 map at batch 2: 200 pixels) runs on the rebuilt tensors through the generic kernel."""
    # w_bt: B is a 1x1 weight used as [k][n] (a data gradient): if its TRANSPOSED planes exist the caller gets those -- and must then
    # run the product with B as [n][k] (PL_STATS["last_t"] says so)
    PL_STATS["last"] = PL_STATS["last_t"] = False   # (did the product just armed take planes?  read by callers)
    if A is None or B is None or a_slot is not None or b_slot is not None or not planes_on() or k_red % 32 != 0:
        if pl_word(A) is not None or pl_word(B) is not None:
            A, B = unplanes(A), unplanes(B)
        h2_arm(A, B, a_slot, b_slot)
        return P(A), P(B)
    wa, wb = pl_word(A), pl_word(B)
    if wa is not None or wb is not None:
        pa = (P(A), wa) if wa is not None else (_pl_weight(A) if w_a else None)
        pb = (P(B), wb) if wb is not None else (_pl_weight(B) if w_b else None)
        if w_bt and wb is None and pa is not None:
            tt = getattr(B, "_plwt", None)
            if tt is not None and tt[0] == _H2["step"]:
                pb = (tt[1], tt[2])
                PL_STATS["last_t"] = True
        if pa is not None and pb is not None:
            call("tris_h2_next_planes", pa[1], pb[1], (1 if y_planes else 0) | (2 if PL_STATS["last_t"] else 0) | (4 if y_mask else 0))
            PL_STATS["products"] += 1
            PL_STATS["last"] = True
            return pa[0], pb[0]
        PL_STATS["mixed"] += 1
        A, B = unplanes(A), unplanes(B)
    h2_arm(A, B)
    return P(A), P(B)


class _BnBwdLink:
    """Hand-off between a train-mode BatchNorm(+ReLU) and the 1x1 convolution / Linear that consumes its output.

    The BatchNorm backward needs sum(dz), sum(dz * xhat) over the whole batch before it can form dx -- a reduction pass over dy
    and x of its own (col_partial_kernel<1>).  Where dy is produced by a 1x1-convolution data gradient, that GEMM's epilogue has
    every dy value in registers: tris_gemm_bnbwd_f32 masks it there, stores the masked gradient dz and leaves the two sums as
    per-tile fp64 partial rows, and the BatchNorm backward is the apply pass alone.  Forward: BatchNormFn.forward hangs a link on
    its output tensor, ops.linear picks it up from its input.  Backward: LinearFn.backward fills it (dz tensor, partial rows),
    BatchNormFn.backward uses it iff the gradient it is handed IS that tensor, unmodified (same storage, same version counter:
    a second consumer's gradient accumulated into it by autograd changes one of the two) -- otherwise the usual two passes run
    on whatever arrived, which is why the masked gradient is only ever produced when the BatchNorm is known to be behind it.
    The model marks the BatchNorms whose output has exactly one autograd consumer (Bottleneck: bn2 -> conv3; bn3 -> the next
    block's conv1, the residual branch rides a GradBox) with bwd_link=True."""
    __slots__ = ("x", "mean", "invstd", "gamma", "beta", "from_y", "dz", "part", "rows", "dzw", "mask")

    def __init__(self, x, mean, invstd, gamma, beta, from_y):
        # (no reference to the BatchNorm's OUTPUT, which carries this object: the consumer has that tensor as its own input)
        self.x, self.mean, self.invstd, self.gamma, self.beta, self.from_y = x, mean, invstd, gamma, beta, from_y
        self.dz = self.part = self.dzw = None
        self.rows = 0
        self.mask = None      # (cfg.bn_bitmask) the ReLU mask of the BatchNorm's plane output, one byte per 8 channels (tris_bn_mask_next)

    def fill(self, dz, part, rows, dzw=None):
        # (identity of the gradient tensor, not a reference to it: autograd hands a sole-owner gradient on without a copy)
        # dzw: amax word of the masked gradient (operand planes: the bound of the BatchNorm's dx is formed from it)
        self.dz, self.part, self.rows, self.dzw = (dz.data_ptr(), dz._version, tuple(dz.shape)), part, rows, dzw

    def take(self, dy):
        """(part, rows) if dy IS the masked gradient this link's product left (same storage, version and shape); None if no product
        filled the link -- or if what arrives is something else: a second consumer's gradient was added to it by autograd (a
        down-sampling block whose shortcut backward ran late, a model that uses the BatchNorm output twice).  The caller then runs
        the usual two passes on whatever arrived, which is correct for any sum of gradients (the ReLU mask is idempotent: masking
        an already-masked term again changes nothing).  One use."""
        dz, part, rows, dzw = self.dz, self.part, self.rows, self.dzw
        self.dz = self.part = self.dzw = None
        if dz is None or (dy.data_ptr(), dy._version, tuple(dy.shape)) != dz:
            return None
        return part, rows, dzw


def _bn_bwd_fuse_enabled():
    return cfg.bn_bwd_fuse


class LinearFn(torch.autograd.Function):
    """y = act(x . W^T + b) + resid  with W [N, K] (nn.Linear) or [N, K, 1, 1] (1x1 conv on channels-last)."""

    @staticmethod
    def forward(ctx, x, w, b, resid, act, stats=False, grad_box=None, grad_box_out=None, grad_box_res=None, bn_link=None,
                act_link=None):
        _chk(x, w, b, resid)
        ctx.grad_box, ctx.grad_box_out, ctx.grad_box_res = grad_box, grad_box_out, grad_box_res
        ctx.act_link = act_link
        ctx.bn_link = bn_link if (bn_link is not None and x.is_contiguous() and bn_link.x.shape == x.shape) else None
        x = x.contiguous()
        K = x.shape[-1]
        N = w.numel() // K
        M = x.numel() // K
        y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        if resid is not None:
            resid = resid.contiguous()
        want_stats = stats and b is None and resid is None and act == 0
        if want_stats:
            _launch_with_stats(y, M, N, lambda part, rows: _timed(
                "gemm", 2.0 * M * N * K, lambda: (lambda pp: call("tris_gemm_bnstat_f32", pp[0], pp[1], P(y), M, N, K,
                                                                  part.data_ptr(), rows, _stream()))(h2_pp(x, w, w_b=True, k_red=K)),
                nbytes=4.0 * (M * K + K * N + M * N), tag=_shape_tag(f"bnstat M{M} N{N} K{K}")))
        else:
            gemm(x, w, y, M, N, K, K, K, N, False, True, bias=b, bias_mode=1 if b is not None else 0, resid=resid,
                 ldr=N, act=act, w_b=True)
        ctx.act, ctx.dims = act, (M, N, K)
        ctx.has_b, ctx.has_r = b is not None, resid is not None
        ctx.params = (w, b)
        ctx.x_pl = getattr(x, "_pl", None)
        ctx.save_for_backward(x, w, b, y if act == 1 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.act == 2:
            raise RuntimeError("fused QuickGELU epilogue is forward-only; use QGeluFn when gradients are needed")
        if ctx.act == 1 and ctx.has_r:
            raise RuntimeError("relu + residual epilogue is forward-only")
        x, w, b, y = ctx.saved_tensors
        pw, pb = ctx.params
        M, N, K = ctx.dims
        dy = pl_grad_in(dy.contiguous())
        _retag(x, ctx.x_pl)
        if pl_word(dy) is not None and (ctx.has_r or ctx.has_b or ctx.act == 1):
            dy = unplanes(dy)      # (bias / residual / ReLU backward read the gradient element-wise: never on the trunk's path)
        if pl_word(dy) is None:
            x = unplanes(x)        # (a plane input next to an fp32 gradient -- vis_project: the weight gradient reads the rebuilt tensor)
        d_res = dy if ctx.has_r and ctx.needs_input_grad[3] else None
        if d_res is not None and ctx.grad_box_res is not None and ctx.grad_box_res.deposit(d_res):
            d_res = None   # picked up by the block's LayerNorm backward (read-only there: no copy needed)
        if d_res is not None and _wgrad_enabled() and ctx.needs_input_grad[1]:
            # dy is about to be read by the weight-gradient stream; autograd may accumulate IN PLACE into a gradient
            # tensor it is handed back (InputBuffer steals sole-owner tensors), so the residual branch gets its own copy
            d_res = dy.clone()
        if ctx.act == 1:
            dy = ew("TRIS_EW_RELU_BWD", dy, y)
        dx = None
        box = ctx.grad_box
        extra = None   # a residual-branch gradient left by a later layer of the same block: added in the GEMM epilogue
        if box is not None:
            box.consumed = True
            if box.value is not None:
                extra, box.value = box.value, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            link, fused = ctx.bn_link, False
            if link is not None and (box is None or extra is not None):
                # x is the output of a BatchNorm(+ReLU) that has no other consumer: mask dx and reduce it for that BatchNorm's
                # backward in this product's epilogue (_BnBwdLink).  With a gradient box whose value has NOT arrived yet (the
                # other consumer's backward is still to run and will hand its term to autograd) the sums would miss that term:
                # the plain data gradient runs and the BatchNorm backward does its own reduction
                import ctypes
                part = torch.empty(((M + 127) // 128) * 2 * K, device=x.device, dtype=torch.float64)
                rows = ctypes.c_int(0)
                ypl = link.from_y and pl_word(x) is not None   # (the BatchNorm's output as planes: only a plane product masks from it)
                if not ypl or pl_word(dy) is not None:
                    dzw = _h2_slot() if planes_on() else None   # (the masked gradient's amax: the bound of that BatchNorm's dx)

                    def launch():
                        bits = ypl and link.mask is not None
                        pp = h2_pp(dy, w, w_b=True, y_planes=ypl, k_red=N, w_bt=True, y_mask=bits)
                        if dzw is not None:
                            call("tris_amax_next", dzw)
                        by = (P(x) if (PL_STATS["last"] or not ypl) else P(unplanes(x))) if link.from_y else None
                        if bits and PL_STATS["last"]:
                            by = link.mask.data_ptr()       # (the product took planes and was armed with flag 4: bn_y is the byte mask)
                        call("tris_gemm_bnbwd_f32", pp[0], pp[1], P(dx), M, K, N, P(extra), K, P(link.x), by,
                             P(link.mean), P(link.invstd), P(link.gamma), P(link.beta), part.data_ptr(), ctypes.byref(rows), _stream())
                        return rows.value > 0     # (False: the entry point declined the shape, nothing was launched)
                    _timed("gemm_bnbwd", 2.0 * M * N * K, launch, tag=_shape_tag(f"M{M} N{K} K{N}"),
                           nbytes=4.0 * (M * N + N * K + M * K * (2 + int(link.from_y) + int(extra is not None))))
                    if rows.value > 0:
                        fused = True
                        link.fill(dx, part, rows.value, dzw)
            alink = ctx.act_link
            if not fused and alink is not None and extra is None:
                # x = QuickGELU(pre) with this product as its only consumer (linear_qgelu): the data gradient comes out of the
                # epilogue already multiplied by quickgelu'(pre); the producer's backward is told so
                if gemm(dy, w, dx, M, K, N, N, K, K, False, False, dact=alink.pre) is not False:
                    fused = alink.applied = True
            if not fused:
                gemm(dy, w, dx, M, K, N, N, K, K, False, False, resid=extra, ldr=K, w_b=True, w_bt=True)
        elif extra is not None:
            raise RuntimeError("a residual gradient was handed to a layer whose input needs no gradient")
        if dx is not None and ctx.grad_box_out is not None and ctx.grad_box_out.deposit(dx):
            dx = None   # the block's first conv adds it in its data-gradient epilogue
        dw = None
        if ctx.needs_input_grad[1]:
            if _sink(pw) is not None:   # into the gradient arena, on the weight-gradient stream
                on_wgrad_stream(lambda: gemm(dy, x, _sink(pw), N, K, M, N, K, K, True, False, mark=False), dy, x, sink=_sink(pw))
            else:
                dw = _emit(pw, lambda o: gemm(dy, x, o, N, K, M, N, K, K, True, False, mark=False), True)
        db = None
        if ctx.has_b:
            db = _emit(pb, lambda o: colsum(dy, M, N, o), ctx.needs_input_grad[2], reads=(dy,))
        return dx, dw, db, d_res, None, None, None, None, None, None, None


def linear(x, w, b=None, resid=None, act=0, stats=False, grad_box=None, grad_box_out=None, grad_box_res=None, act_link=False):
    """act_link=True: x is the output of linear_qgelu and THIS product is its only consumer -- the QuickGELU backward rides this
    product's data-gradient epilogue (the caller vouches for the single consumer)"""
    link = getattr(x, "_bn_link", None)
    if link is not None and not (torch.is_grad_enabled() and x.requires_grad and _bn_bwd_fuse_enabled()):
        link = None
    alink = getattr(x, "_act_link", None) if (act_link and torch.is_grad_enabled() and x.requires_grad) else None
    return LinearFn.apply(x, w, b, resid, act, stats, grad_box, grad_box_out, grad_box_res, link, alink)


class _ActLink:
    """QuickGELU hand-off between linear_qgelu (producer: holds the pre-activation) and the Linear that consumes its output:
    `applied` is set by the consumer's backward when its data gradient already carries quickgelu'(pre)."""
    __slots__ = ("pre", "applied")

    def __init__(self, pre):
        self.pre, self.applied = pre, False


class LinearQGeluFn(torch.autograd.Function):
    """QuickGELU(x . W^T + b) in one launch (CLIP/clip/model.py:361-376: c_fc + gelu of the transformer MLP): the epilogue stores the
    activated value AND the pre-activation its backward needs."""

    @staticmethod
    def forward(ctx, x, w, b):
        _chk(x, w, b)
        x = x.contiguous()
        K = x.shape[-1]
        N = w.numel() // K
        M = x.numel() // K
        y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        pre = torch.empty_like(y)
        if gemm(x, w, y, M, N, K, K, K, N, False, True, bias=b, bias_mode=1 if b is not None else 0, act=2, pre_out=pre) is False:
            gemm(x, w, pre, M, N, K, K, K, N, False, True, bias=b, bias_mode=1 if b is not None else 0)
            ew("TRIS_EW_QGELU", pre, out=y)
        ctx.dims, ctx.has_b, ctx.params = (M, N, K), b is not None, (w, b)
        ctx.link = y._act_link = _ActLink(pre)
        ctx.save_for_backward(x, w, pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, pre = ctx.saved_tensors
        pw, pb = ctx.params
        M, N, K = ctx.dims
        dy = dy.contiguous()
        dpre = dy if ctx.link.applied else ew("TRIS_EW_QGELU_BWD", dy, pre)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(dpre, w, dx, M, K, N, N, K, K, False, False)
        dw = None
        if ctx.needs_input_grad[1]:
            if _sink(pw) is not None:
                on_wgrad_stream(lambda: gemm(dpre, x, _sink(pw), N, K, M, N, K, K, True, False, mark=False), dpre, x, sink=_sink(pw))
            else:
                dw = _emit(pw, lambda o: gemm(dpre, x, o, N, K, M, N, K, K, True, False, mark=False), True)
        db = None
        if ctx.has_b:
            db = _emit(pb, lambda o: colsum(dpre, M, N, o), ctx.needs_input_grad[2], reads=(dpre,))
        return dx, dw, db


def linear_qgelu(x, w, b=None):
    return LinearQGeluFn.apply(x, w, b)     # (the returned tensor carries `_act_link`, set inside forward)


class MatmulFn(torch.autograd.Function):
    """C = A . B (tB False, B [K,N]) or A . B^T (tB True, B [N,K]); A [..., K] flattened to 2-D."""

    @staticmethod
    def forward(ctx, A, B, tB):
        _chk(A, B)
        A, B = A.contiguous(), B.contiguous()
        K = A.shape[-1]
        M = A.numel() // K
        N = B.shape[0] if tB else B.shape[1]
        C = torch.empty(*A.shape[:-1], N, device=A.device, dtype=torch.float32)
        gemm(A, B, C, M, N, K, K, B.shape[1], N, False, tB)
        ctx.tB, ctx.dims = tB, (M, N, K)
        ctx.params = (B,)
        ctx.save_for_backward(A, B)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        M, N, K = ctx.dims
        dC = dC.contiguous()
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty_like(A)
            if ctx.tB:   # dA = dC . B   (B [N,K])
                gemm(dC, B, dA, M, K, N, N, K, K, False, False)
            else:        # dA = dC . B^T (B [K,N])
                gemm(dC, B, dA, M, K, N, N, N, K, False, True)
        if ctx.needs_input_grad[1]:
            if ctx.tB:   # dB [N,K] = dC^T . A
                dB = _emit(ctx.params[0], lambda o: gemm(dC, A, o, N, K, M, N, K, K, True, False, mark=False), True)
            else:        # dB [K,N] = A^T . dC
                dB = _emit(ctx.params[0], lambda o: gemm(A, dC, o, K, N, M, K, N, N, True, False, mark=False), True)
        return dA, dB, None


def matmul(A, B, tB=False):
    return MatmulFn.apply(A, B, tB)


class BmmFn(torch.autograd.Function):
    """C[b] = A[b] . op(B[b]);  A may be 2-D (shared by every batch).  A [Bt,M,K]|[M,K]; B [Bt,N,K] (tB) | [Bt,K,N]."""

    @staticmethod
    def forward(ctx, A, B, tB, alpha):
        _chk(A, B)
        A, B = A.contiguous(), B.contiguous()
        Bt = B.shape[0]
        shared = A.dim() == 2
        M, K = A.shape[-2], A.shape[-1]
        N = B.shape[1] if tB else B.shape[2]
        C = torch.empty(Bt, M, N, device=B.device, dtype=torch.float32)
        gemm(A, B, C, M, N, K, K, B.shape[2], N, False, tB, batch=Bt, sA=0 if shared else M * K,
             sB=B.shape[1] * B.shape[2], sC=M * N, alpha=alpha)
        ctx.cfg = (tB, alpha, shared, Bt, M, N, K)
        ctx.save_for_backward(A, B)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        tB, alpha, shared, Bt, M, N, K = ctx.cfg
        dC = dC.contiguous()
        sA = 0 if shared else M * K
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dAp = torch.empty(Bt, M, K, device=B.device, dtype=torch.float32)
            if tB:   # dA[b] = dC[b] . B[b]       (B[b] is [N,K])
                gemm(dC, B, dAp, M, K, N, N, K, K, False, False, batch=Bt, sA=M * N, sB=N * K, sC=M * K, alpha=alpha)
            else:    # dA[b] = dC[b] . B[b]^T     (B[b] is [K,N])
                gemm(dC, B, dAp, M, K, N, N, N, K, False, True, batch=Bt, sA=M * N, sB=K * N, sC=M * K, alpha=alpha)
            if shared:
                dA = torch.empty(M, K, device=B.device, dtype=torch.float32)
                colsum(dAp, Bt, M * K, dA)
            else:
                dA = dAp
        if ctx.needs_input_grad[1]:
            dB = torch.empty_like(B)
            if tB:   # dB[b] [N,K] = dC[b]^T . A[b]
                gemm(dC, A, dB, N, K, M, N, K, K, True, False, batch=Bt, sA=M * N, sB=sA, sC=N * K, alpha=alpha)
            else:    # dB[b] [K,N] = A[b]^T . dC[b]
                gemm(A, dC, dB, K, N, M, K, N, N, True, False, batch=Bt, sA=sA, sB=M * N, sC=K * N, alpha=alpha)
        return dA, dB, None, None


def bmm(A, B, tB=False, alpha=1.0):
    return BmmFn.apply(A, B, tB, alpha)


# ----------------------------------------------------------------------------------------------- 3x3 conv
class Conv3x3Fn(torch.autograd.Function):
    """x [B,H,W,Cin] channels-last, w [Cout,Cin,3,3] in channels_last memory, pad 1, stride 1|2."""

    @staticmethod
    def forward(ctx, x, w, stride, stats=False, lazy=None, bn_link=None):
        """bn_link: x is the output of a BatchNorm + ReLU with no other consumer (_BnBwdLink): the data gradient reduces that
        BatchNorm's backward sums in its epilogue.
        lazy = (x_raw, mean, invstd, gamma, beta): `x` is the NOT-YET-WRITTEN output buffer of a BatchNorm + ReLU over x_raw
        (batch_norm(..., lazy=True)); the direct kernels normalise x_raw while they stage it and `x` is never written -- or, when
        they cannot serve the shape after all, it is written here first and everything proceeds as usual.  (h2: the unwritten
        buffer carries the amax word of an upper bound of its would-be contents, BatchNormFn.forward.)"""
        _chk(x, w)
        ctx.bn_link = bn_link if (bn_link is not None and stride == 1 and not bn_link.from_y and x.is_contiguous()
                                  and bn_link.x.shape == x.shape) else None
        x = x.contiguous()
        ctx.params = (w,)
        w = cl_weight(w)
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        ctx.lazy = None
        ctx.stride = stride
        if lazy is not None:
            xr, mean, invstd, gamma, beta, bound = lazy
            if stride == 1 and conv3x3_bnin_ok(x.shape, Cout):
                y = torch.empty(B, H, W, Cout, device=x.device, dtype=torch.float32)
                fl = 2.0 * B * H * W * Cout * 9 * Cin
                nb = 4.0 * (B * H * W * (Cin + Cout) + 9 * Cin * Cout)
                ctx.h2_in = bound      # (None outside h2)

                def launch(part, rows):
                    return _timed("conv3x3_fwd", fl, lambda: (bound is not None and h2_arm(None, ctx.params[0], a_slot=bound), call(
                        "tris_conv3x3_fwd_bnin_f32", P(xr), P(mean), P(invstd), P(gamma), P(beta), P(w), P(y), B, H, W, Cin, Cout,
                        None if part is None else part.data_ptr(), rows, _stream()))[1], nbytes=nb,
                        tag=_shape_tag(f"bnin B{B} {H}x{W} {Cin}->{Cout}"))
                if stats:
                    _launch_with_stats(y, B * H * W, Cout, launch)
                else:
                    launch(None, None)
                ctx.lazy = True
                ctx.save_for_backward(xr, w, mean, invstd, gamma, beta)
                return y
            h2_mark_next(x)
            call("tris_bn_apply_f32", P(xr), P(mean), P(invstd), P(gamma), P(beta), None, P(x), B * H * W, Cin, 1, _stream())
        if pl_word(x) is not None and (Cin % 32 != 0 or stride != 1):
            x = unplanes(x)        # (the fast kernels' gather needs 32-channel pieces; the stem's strided first convolution reads the image)
        ctx.x_pl = getattr(x, "_pl", None)
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
        fl = 2.0 * B * Ho * Wo * Cout * 9 * Cin
        nb = 4.0 * (B * H * W * Cin + B * Ho * Wo * Cout + 9 * Cin * Cout)
        if stats:
            _launch_with_stats(y, B * Ho * Wo, Cout, lambda part, rows: _timed(
                "conv3x3_fwd", fl, lambda: (lambda pp: call("tris_conv3x3_fwd_bnstat_f32", pp[0], pp[1], P(y), B, H, W, Cin, Cout,
                                                            stride, part.data_ptr(), rows, _stream()))(_conv_pp(x, ctx.params[0], w)),
                nbytes=nb, tag=_shape_tag(f"bnstat B{B} {H}x{W} {Cin}->{Cout} s{stride}")))
        else:
            _timed("conv3x3_fwd", fl,
                   lambda: (lambda pp: call("tris_conv3x3_fwd_f32", pp[0], pp[1], P(y), B, H, W, Cin, Cout, stride, _stream()))(
                       _conv_pp(x, ctx.params[0], w)), nbytes=nb, tag=_shape_tag(f"B{B} {H}x{W} {Cin}->{Cout} s{stride}"))
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.lazy:
            x, w, mean, invstd, gamma, beta = ctx.saved_tensors   # x = the BatchNorm's raw input
        else:
            x, w = ctx.saved_tensors
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        dy = pl_grad_in(dy.contiguous())
        if not ctx.lazy:
            _retag(x, ctx.x_pl)
        if pl_word(dy) is not None and (ctx.lazy or pl_word(x) is None or Cout % 32 != 0):
            dy = unplanes(dy)      # (the BatchNorm-folded forms and the stem's first convolution read fp32)
        if pl_word(dy) is None:
            x = unplanes(x)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.stride != 1:
                raise NotImplementedError("dgrad of the strided stem conv is never needed (its input is the image)")
            dx = torch.empty_like(x)
            fl = 2.0 * B * H * W * Cout * 9 * Cin
            link, fused = ctx.bn_link, False
            if link is not None:   # (see LinearFn.backward)
                import ctypes
                part = torch.empty(((B * H * W + 127) // 128) * 2 * Cin, device=x.device, dtype=torch.float64)
                rows = ctypes.c_int(0)
                dzw = _h2_slot() if planes_on() else None   # (the masked gradient's amax: the bound of that BatchNorm's dx)

                def launch():
                    pp = _conv_pp(dy, ctx.params[0], w)
                    if dzw is not None:
                        call("tris_amax_next", dzw)
                    call("tris_conv3x3_dgrad_bnbwd_f32", pp[0], pp[1], P(dx), B, H, W, Cin, Cout, P(link.x), P(link.mean), P(link.invstd),
                         P(link.gamma), P(link.beta), part.data_ptr(), ctypes.byref(rows), _stream())
                    return rows.value > 0
                _timed("conv3x3_dgrad_bnbwd", fl, launch, tag=_shape_tag(f"B{B} {H}x{W} {Cin}<-{Cout}"), nbytes=4.0 * (B * H * W * (Cout + 2 * Cin) + 9 * Cin * Cout))
                if rows.value > 0:
                    fused = True
                    link.fill(dx, part, rows.value, dzw)
            if not fused:
                _timed("conv3x3_dgrad", fl,
                       lambda: (lambda pp: call("tris_conv3x3_dgrad_f32", pp[0], pp[1], P(dx), B, H, W, Cin, Cout, _stream()))(
                           _conv_pp(dy, ctx.params[0], w)), tag=_shape_tag(f"B{B} {H}x{W} {Cin}<-{Cout}"),
                       nbytes=4.0 * (B * H * W * (Cout + Cin) + 9 * Cin * Cout))

        def wgrad(o):
            ws = workspace(0)
            fl = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * Cout * 9 * Cin
            nb = 4.0 * (B * H * W * Cin + dy.numel() + 9 * Cin * Cout)
            xin = x
            if ctx.lazy:
                if conv3x3_bnin_ok(x.shape, Cout):
                    return _timed("conv3x3_wgrad", fl, lambda: (ctx.h2_in is not None and h2_arm(dy, None, b_slot=ctx.h2_in), call(
                        "tris_conv3x3_wgrad_bnin_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), P(dy), P(o), B, H, W, Cin,
                        Cout, P(ws), ws.numel() * 4, _stream()))[1], nbytes=nb, tag=_shape_tag(f"bnin B{B} {H}x{W} {Cin}->{Cout}"))
                xin = torch.empty_like(x)   # materialise relu(bn(x)) after all
                h2_mark_next(xin)
                call("tris_bn_apply_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), None, P(xin), B * H * W, Cin, 1, _stream())
            return _timed("conv3x3_wgrad", fl, lambda: (lambda pp: call(
                "tris_conv3x3_wgrad_f32", pp[1], pp[0], P(o), B, H, W, Cin, Cout, ctx.stride, P(ws), ws.numel() * 4, _stream()))(
                    h2_pp(dy, xin, k_red=dy.shape[0] * dy.shape[1] * dy.shape[2])), nbytes=nb,
                tag=_shape_tag(f"B{B} {H}x{W} {Cin}->{Cout} s{ctx.stride}"))
        dw = None
        if ctx.needs_input_grad[1]:
            sk = _sink(ctx.params[0])
            if sk is not None:
                # (every tensor the deferred launch READS is named: a segmented capture keeps them alive and keeps the early part
                # of AdamW away from the parameters among them -- the folded BatchNorm's gamma / beta)
                on_wgrad_stream(lambda: wgrad(sk), dy, x, *((mean, invstd, gamma, beta) if ctx.lazy else ()), sink=sk)
            else:
                dw = _emit(ctx.params[0], wgrad, True)
        return dx, dw, None, None, None, None


def _conv_pp(a, w_param, w_cl):
    """operand pointers of a 3x3 product (activation or gradient `a`, weight parameter `w_param` whose kernel layout is `w_cl`)"""
    pa, pw = h2_pp(a, w_param, w_b=True)
    return pa, (pw if pw != P(w_param) else P(w_cl))


def conv3x3_bnin_ok(xshape, Cout):
    """can conv3x3(relu(bn(x))) run with the BatchNorm folded into the direct kernels (forward AND weight gradient)?"""
    B, H, W, Cin = xshape
    return bool(query("tris_conv3x3_bnin_ok", B, H, W, Cin, Cout))


def conv3x3(x, w, stride=1, stats=False):
    link = getattr(x, "_bn_link", None)
    if link is not None and not (torch.is_grad_enabled() and x.requires_grad and _bn_bwd_fuse_enabled()):
        link = None
    return Conv3x3Fn.apply(x, w, stride, stats, getattr(x, "_bn_lazy", None), link)


# ----------------------------------------------------------------------------------------------- BatchNorm
class BatchNormFn(torch.autograd.Function):
    """BatchNorm2d on channels-last x [..., C] with optional fused residual add and ReLU.

    training=True: batch statistics (+ running-stat update, + cross-rank combine when `group` is given =
    SyncBatchNorm).  training=False: running statistics (forward only)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, resid, relu, training, momentum, eps, group, part=None, grad_box=None,
                lazy=False, bwd_link=False, pool=False, planes=False, dx_planes=False):
        """planes: the output is WRITTEN as fp16 operand planes (every consumer is a product, a plane-aware BatchNorm / pool, or goes
        through ops.unplanes); dx_planes: so is the input gradient (its one consumer is the backward of the convolution before)."""
        _chk(x, gamma, beta, rmean, rvar, resid)
        y_mask = None
        x = x.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        use_pl = bool(planes and training and C % 8 == 0 and 2048 % C == 0 and planes_on())
        ctx.dx_pl = bool(dx_planes and training and C % 8 == 0 and 2048 % C == 0 and planes_on())
        if use_pl:
            lazy = False
        # pool: the output is avgpool2(relu(bn(x))) -- BatchNorm + ReLU + AvgPool2d(2) as one op, the full-size tensor never written
        pool = bool(pool)   # (ops.batch_norm has checked: train mode, ReLU, no residual, even map)
        y = (torch.empty(x.shape[0], x.shape[1] // 2, x.shape[2] // 2, C, device=x.device, dtype=torch.float32) if pool
             else torch.empty_like(x))
        if resid is not None:
            resid = resid.contiguous()
        count = M
        if training:
            stats = torch.empty(3 * C, device=x.device, dtype=torch.float32)

            pl_bound = {"word": None, "resid": None}   # (operand planes: the output's bound word, formed by the finalizer where it can)

            def local_stats(rm, rv, dst=None):
                dst = stats if dst is None else dst
                if part is not None and use_pl and group is None:
                    # partial sums from the producing conv's epilogue; the same launch leaves the bound of the plane output
                    w_, rw_ = _h2_slot(), None
                    if resid is not None:
                        rw_ = pl_word(resid)
                        if rw_ is None:
                            rw_ = _h2_amax(resid)
                    if w_ is not None and (resid is None or rw_ is not None):
                        call("tris_bn_finalize_bound_f32", part[0].data_ptr(), part[1], M, C, eps, momentum, P(dst), P(rm), P(rv),
                             P(gamma), P(beta), math.sqrt(max(M - 1, 1)), rw_, w_, _stream())
                        pl_bound["word"], pl_bound["resid"] = w_, rw_
                        return
                if part is not None:  # partial sums came out of the producing conv's epilogue
                    call("tris_bn_finalize_f32", part[0].data_ptr(), part[1], M, C, eps, momentum, P(dst), P(rm),
                         P(rv), _stream())
                else:
                    ws = workspace(query("tris_col_workspace_bytes", M, C))
                    call("tris_bn_stats_f32", P(x), M, C, eps, momentum, P(dst), P(rm), P(rv), P(ws), _stream())
            if group is None:
                local_stats(rmean, rvar)
            else:
                from . import comm
                world = comm.world_size(group)
                mb = comm.syncbn_mailbox(group, 3 * C)
                if mb is not None:   # one launch: peer-mailbox exchange + combine (+ running statistics)
                    loc = torch.empty(3 * C, device=x.device, dtype=torch.float32)
                    local_stats(None, None, dst=loc)
                    w_ = rw_ = None
                    if use_pl and cfg.syncbn_bound:   # ... + the bound word of the plane output (what tris_bn_out_bound2_f32 would launch for)
                        w_ = _h2_slot()
                        if resid is not None:
                            rw_ = pl_word(resid)
                            if rw_ is None:
                                rw_ = _h2_amax(resid)
                    if w_ is not None and (resid is None or rw_ is not None):
                        mb.bn_combine_bound(loc, C, M, eps, momentum, stats, rmean, rvar, gamma, beta,
                                            math.sqrt(max(M * world - 1, 1)), rw_, w_)
                        pl_bound["word"], pl_bound["resid"] = w_, rw_
                    else:
                        mb.bn_combine(loc, C, M, eps, momentum, stats, rmean, rvar)
                else:
                    local_stats(None, None)
                    allv = torch.empty(world * 3 * C, device=x.device, dtype=torch.float32)
                    comm.syncbn_all_gather_into(allv, stats, group=group)   # the stats block travels as is
                    call("tris_bn_sync_combine_f32", P(allv), world, C, M, eps, momentum, P(stats), P(rmean), P(rvar),
                         _stream())
                count = M * world  # DistributedSampler gives every rank the same per-step batch
            mean, invstd = stats[:C], stats[C:2 * C]
        else:
            mean = rmean
            invstd = torch.rsqrt(rvar + eps)
        # lazy: y stays UNWRITTEN -- its only consumer (a 3x3 convolution with direct kernels) normalises x while staging it
        lazy = bool(lazy and training and relu and resid is None)
        if not use_pl and resid is not None:
            resid = unplanes(resid)
        if use_pl:
            # the scale must exist before the pass writes: Samuelson's bound of the normalised values from the affine parameters,
            # plus the bound (or amax) of the residual it is added to
            rk = 0 if resid is None else (2 if pl_word(resid) is not None else 1)
            word = pl_bound["word"] if training else None
            rw = pl_bound["resid"] if word is not None else None
            if word is None:
                word = _h2_slot()
                if resid is not None:
                    rw = pl_word(resid)
                    if rw is None:
                        rw = _h2_amax(resid)
                if word is None or (resid is not None and rw is None):
                    raise RuntimeError("operand planes: no amax word for a BatchNorm output / its residual")
                call("tris_bn_out_bound2_f32", P(gamma), P(beta), C, math.sqrt(max(count - 1, 1)), rw, word, _stream())
            if pool:
                call("tris_bn_apply_pool_pl_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), P(y), word, x.shape[0], x.shape[1],
                     x.shape[2], C, _stream())
            else:
                if cfg.bn_bitmask and training and bwd_link and relu and resid is not None:
                    # out = relu(bn(x) + identity) with ONE consumer whose data-gradient epilogue will mask with it (_BnBwdLink.from_y):
                    # the pass also leaves the mask as one byte per 8 channels, so that epilogue reads a bit, not the plane element
                    y_mask = torch.empty(M * C // 8, device=x.device, dtype=torch.uint8)
                    call("tris_bn_mask_next", y_mask.data_ptr())
                call("tris_bn_apply_pl_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), P(resid), rk, rw if rk == 2 else None, P(y),
                     word, M, C, int(relu), _stream())
            pl_tag(y, word)
        elif pool:
            h2_mark_next(y)
            call("tris_bn_apply_pool_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), P(y), x.shape[0], x.shape[1], x.shape[2], C,
                 _stream())
        elif not lazy:
            h2_mark_next(y)
            call("tris_bn_apply_f32", P(x), P(mean), P(invstd), P(gamma), P(beta), P(resid), P(y), M, C, int(relu),
                 _stream())
        if lazy:   # the hand-off to ops.conv3x3 rides on the (unwritten) output tensor itself
            # h2: the consuming convolution needs an operand scale for a tensor that never exists -- an upper bound of |y| from the
            # affine parameters alone (Samuelson: |xhat| <= sqrt(n - 1)), as an amax word that travels with the hand-off
            y._bn_lazy = (x, mean, invstd, gamma, beta, h2_bound_word(gamma, beta, C, math.sqrt(max(count - 1, 1))))
        ctx.cfg = (M, C, bool(relu), resid is not None, count, group)
        ctx.grad_box = grad_box
        ctx.params = (gamma, beta)
        ctx.training = bool(training)
        ctx.link = None
        ctx.pool = pool
        ctx.y_pl = getattr(y, "_pl", None)
        if training:
            keep_y = relu and not pool and resid is not None
            ctx.save_for_backward(x, gamma, beta, mean, invstd, y if keep_y else None)
            if bwd_link and relu and not pool:
                # the consumer (ops.linear) may reduce this BatchNorm's backward sums in its data-gradient epilogue: _BnBwdLink
                ctx.link = y._bn_link = _BnBwdLink(x, mean, invstd, gamma, beta, keep_y)
                ctx.link.mask = y_mask if keep_y else None
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise NotImplementedError("BatchNorm in eval mode is forward-only on this path")
        x, gamma, beta, mean, invstd, y = ctx.saved_tensors
        M, C, relu, has_res, count, group = ctx.cfg
        dy = unplanes(pl_grad_in(dy.contiguous()))   # (the gradient of a BatchNorm OUTPUT is fp32 on every path of the trunk)
        _retag(y, ctx.y_pl)
        y_is_pl = pl_word(y) is not None
        dx_pl = ctx.dx_pl and planes_on()
        dzw = _h2_slot() if dx_pl else None    # amax word of the masked gradient dz: left by whichever pass reduces it
        d_res = None
        want_dz = has_res and ctx.needs_input_grad[5]
        if has_res and not relu:
            d_res, want_dz = (dy if want_dz else None), False
        # out = relu(bn(x) + resid): both branches see dz = dy * (out > 0); the mask is applied inside the reduce / apply
        # kernels and the apply kernel emits dz for the residual branch in the same pass
        ws = workspace(query("tris_col_workspace_bytes", M, C))
        sg, sb = _sink(ctx.params[0]), _sink(ctx.params[1])
        in_arena = (sg is not None and sb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
                    and sg.is_contiguous() and sb.is_contiguous())
        mb = None
        if group is not None and in_arena:
            from . import comm
            mb = comm.syncbn_mailbox(group, 2 * C)
        direct = in_arena and (group is None or mb is not None)
        if direct:   # the two LOCAL reductions ARE dbeta / dgamma: write them straight into the gradient arena
            p_dz, p_dzx = P(sb), P(sg)
            dg = db = None
        else:
            sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
            p_dz, p_dzx = P(sums), P(sums, C)
        # BatchNorm + ReLU without a residual: the mask is recomputed from x inside the kernels, y is not read
        mask_x = relu and not has_res and y is None
        # residual blocks (out = relu(bn(x) + identity)): the reduce pass writes the masked gradient dz -- which IS the identity
        # branch's gradient -- and the apply pass reads it back instead of masking dy from y a second time
        got = ctx.link.take(dy) if ctx.link is not None else None
        if got is not None and dx_pl:
            dzw = got[2] if got[2] is not None else _h2_amax(dy)
        if ctx.pool:   # dy is the POOLED gradient: both passes read it in place of a full-size tensor (mask from x)
            Bn, H, W = x.shape[0], x.shape[1], x.shape[2]
            if dzw is not None:
                call("tris_amax_next", dzw)
            call("tris_bn_bwd_reduce_pool_f32", P(dy), P(x), P(mean), P(invstd), Bn, H, W, C, p_dz, p_dzx, P(ws), P(gamma), P(beta),
                 _stream())
        dz_first = got is None and not ctx.pool and want_dz and relu and y is not None
        if dz_first:
            d_res = torch.empty_like(x)
        bound_word = None     # (operand planes: dx's bound word when the finalizer below formed it)
        if got is not None and dx_pl and mb is None and (direct or group is None) and ctx.needs_input_grad[0]:
            bound_word = _h2_slot()
        if got is not None and bound_word is not None:
            call("tris_part_finalize_bound_f32", got[0].data_ptr(), got[1], C, p_dz, p_dzx, P(gamma), P(invstd), 1.0 / float(count),
                 math.sqrt(max(count - 1, 1)), dzw, bound_word, _stream())
        elif got is not None:
            # dy came out of the consuming 1x1 convolution's data gradient already MASKED, with the two sums as partial rows
            call("tris_part_finalize_f32", got[0].data_ptr(), got[1], C, p_dz, p_dzx, _stream())
        elif not ctx.pool:
            if dzw is not None:
                call("tris_amax_next", dzw)
            if y_is_pl and not mask_x:
                call("tris_bn_bwd_reduce_pl_f32", P(dy), P(y), P(x), P(mean), P(invstd), M, C, p_dz, p_dzx, P(ws),
                     P(d_res) if dz_first else None, _stream())
            else:
                call("tris_bn_bwd_reduce_f32", P(dy), None if mask_x else P(y), P(x), P(mean), P(invstd), M, C, p_dz, p_dzx, P(ws),
                     P(gamma) if mask_x else None, P(beta) if mask_x else None, P(d_res) if dz_first else None, _stream())
        if mb is not None:
            # SyncBatchNorm: the arena keeps this rank's dbeta / dgamma (the data-parallel reducer averages them like every
            # other gradient); the sums over ALL ranks that dX needs come from one peer-mailbox launch reading the arena
            sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
            if dx_pl and dzw is not None and ctx.needs_input_grad[0] and bound_word is None and cfg.syncbn_bound:
                bound_word = _h2_slot()
            if bound_word is not None and dx_pl and dzw is not None:   # the same launch leaves the bound word of dx
                mb.bn_bwd_exchange(sb, sg, sums, gamma, invstd, 1.0 / float(count), math.sqrt(max(count - 1, 1)), dzw, bound_word)
            else:
                mb.exchange(sb, sums, 1, src1=sg)
            p_dz, p_dzx = P(sums), P(sums, C)
        elif not direct:
            dg = _emit(ctx.params[0], lambda o: o.copy_(sums[C:]), ctx.needs_input_grad[1])
            db = _emit(ctx.params[1], lambda o: o.copy_(sums[:C]), ctx.needs_input_grad[2])
            if group is not None:
                from . import comm
                comm.syncbn_all_reduce_sum(sums, group=group)
        dx = None
        inv_cnt = 1.0 / float(count)

        def apply(g, ymask, dz_out, beta_m):
            """dx from the (masked or to-be-masked) gradient g; as operand planes where the convolution before takes them"""
            if dx_pl:
                word = bound_word
                if word is None:
                    word = _h2_slot()
                    call("tris_bn_bwd_bound_f32", P(gamma), P(invstd), p_dz, p_dzx, C, inv_cnt, math.sqrt(max(count - 1, 1)), dzw, word,
                         _stream())
                if ctx.pool:
                    call("tris_bn_bwd_apply_pool_pl_f32", P(g), P(x), P(mean), P(invstd), P(gamma), P(beta), p_dz, p_dzx, inv_cnt, P(dx),
                         word, x.shape[0], x.shape[1], x.shape[2], C, _stream())
                else:
                    if ymask is not None and not y_is_pl:
                        raise RuntimeError("operand planes: the ReLU mask of a plane-output BatchNorm must come from plane y")
                    call("tris_bn_bwd_apply_pl_f32", P(g), P(ymask), P(x), P(mean), P(invstd), P(gamma), p_dz, p_dzx, inv_cnt, P(dx),
                         word, P(dz_out), M, C, P(beta_m), _stream())
                pl_grad_out(dx, word)
                return
            h2_mark_next(dx)
            if ctx.pool:
                call("tris_bn_bwd_apply_pool_f32", P(g), P(x), P(mean), P(invstd), P(gamma), P(beta), p_dz, p_dzx, inv_cnt, P(dx),
                     x.shape[0], x.shape[1], x.shape[2], C, _stream())
            else:
                call("tris_bn_bwd_apply_f32", P(g), P(unplanes(ymask)) if ymask is not None else None, P(x), P(mean), P(invstd),
                     P(gamma), p_dz, p_dzx, inv_cnt, P(dx), P(dz_out), M, C, P(beta_m), _stream())
        if ctx.pool:
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                apply(dy, None, None, None)
        elif got is not None:
            if want_dz:
                d_res = dy          # the masked gradient IS the identity branch's gradient
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                apply(dy, None, None, None)
        elif dz_first:
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                apply(d_res, None, None, None)
        elif ctx.needs_input_grad[0] or want_dz:
            dx = torch.empty_like(x)
            if want_dz:
                d_res = torch.empty_like(x)
            apply(dy, None if mask_x else y, d_res if want_dz else None, beta if mask_x else None)
        if ctx.grad_box is not None and d_res is not None and ctx.grad_box.deposit(d_res):
            d_res = None   # handed to the block's first conv
        return dx, dg, db, None, None, d_res, None, None, None, None, None, None, None, None, None, None, None, None


def batch_norm(x, gamma, beta, rmean, rvar, resid=None, relu=False, training=True, momentum=0.1, eps=1e-5, group=None,
               grad_box=None, lazy=False, bwd_link=False, pool=False, planes=False, dx_planes=False):
    """lazy=True (train-mode BatchNorm + ReLU whose ONLY consumer is ops.conv3x3, and conv3x3_bnin_ok said yes): the returned
    tensor is an unwritten buffer carrying `_bn_lazy`; pass it to ops.conv3x3 and nowhere else.
    planes / dx_planes (h2 with operand planes, train mode): the output / the input gradient is written as fp16 operand planes --
    the caller vouches that every consumer takes them (BatchNormFn.forward)."""
    part = getattr(x, "_bn_part", None) if training else None
    bwd_link = bool(bwd_link and training and torch.is_grad_enabled() and x.requires_grad and _bn_bwd_fuse_enabled())
    if pool and not (training and relu and resid is None and not lazy and x.dim() == 4 and x.shape[1] % 2 == 0
                     and x.shape[2] % 2 == 0 and cfg.bn_pool):
        # eval mode / shapes the fused op does not take: the two ops one after the other
        return avgpool2(BatchNormFn.apply(x, gamma, beta, rmean, rvar, resid, relu, training, momentum, eps, group, part, grad_box,
                                          lazy, bwd_link, False, planes, dx_planes))
    return BatchNormFn.apply(x, gamma, beta, rmean, rvar, resid, relu, training, momentum, eps, group, part, grad_box, lazy,
                             bwd_link, pool, planes, dx_planes)


class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grad_box_out=None):
        _chk(x)
        ctx.grad_box_out = grad_box_out
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=torch.float32)
        w = pl_word(x)
        if w is not None and C % 8 == 0:   # planes in, planes out at the same scale (a mean never exceeds the bound of its terms)
            call("tris_avgpool2_fwd_pl_f32", P(x), P(y), w, B, H, W, C, _stream())
            pl_tag(y, w)
        else:
            call("tris_avgpool2_fwd_f32", P(unplanes(x)), P(y), B, H, W, C, _stream())
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dy = unplanes(pl_grad_in(dy.contiguous()))
        dx = torch.empty(B, H, W, C, device=dy.device, dtype=torch.float32)
        call("tris_avgpool2_bwd_f32", P(dy), P(dx), B, H, W, C, _stream())
        if ctx.grad_box_out is not None and ctx.grad_box_out.deposit(dx):
            dx = None
        return dx, None


def avgpool2(x, grad_box_out=None):
    return AvgPool2Fn.apply(x, grad_box_out)


# ----------------------------------------------------------------------------------------------- transformer pieces
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps, grad_box=None):
        _chk(x, g, b)
        ctx.grad_box = grad_box
        x = x.contiguous()
        W = x.shape[-1]
        rows = x.numel() // W
        y = torch.empty_like(x)
        st = torch.empty(2, rows, device=x.device, dtype=torch.float32)
        h2_mark_next(y)
        call("tris_layernorm_fwd_f32", P(x), P(g), P(b), P(y), P(st), P(st, rows), rows, W, eps, _stream())
        ctx.dims = (rows, W)
        ctx.params = (g, b)
        ctx.save_for_backward(x, g, b, st)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, st = ctx.saved_tensors
        rows, W = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        box, extra = ctx.grad_box, None
        if box is not None:   # the block's residual gradient (left by the Linear that added x back): summed in this kernel
            box.consumed = True
            if box.value is not None:
                extra, box.value = box.value.contiguous(), None
                if dx is None:
                    raise RuntimeError("a residual gradient was handed to a LayerNorm whose input needs no gradient")
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        ws = workspace(query("tris_layernorm_bwd_workspace_bytes", rows, W))
        sg, sb = _sink(ctx.params[0]), _sink(ctx.params[1])
        if need_p and sg is not None and sb is not None and sg.is_contiguous() and sb.is_contiguous():
            h2_mark_next(dx)
            call("tris_layernorm_bwd_f32", P(dy), P(x), P(g), P(st), P(st, rows), P(dx), P(sg), P(sb), rows, W, P(ws),
                 P(extra), _stream())
            return dx, None, None, None, None
        dgb = torch.empty(2, W, device=x.device, dtype=torch.float32) if need_p else None
        h2_mark_next(dx)
        call("tris_layernorm_bwd_f32", P(dy), P(x), P(g), P(st), P(st, rows), P(dx), P(dgb), P(dgb, W),
             rows, W, P(ws), P(extra), _stream())
        dg = _emit(ctx.params[0], lambda o: o.copy_(dgb[0]), ctx.needs_input_grad[1])
        db = _emit(ctx.params[1], lambda o: o.copy_(dgb[1]), ctx.needs_input_grad[2])
        return dx, dg, db, None, None


def layer_norm(x, g, b, eps=1e-5, grad_box=None):
    return LayerNormFn.apply(x, g, b, eps, grad_box)


class MhaFn(torch.autograd.Function):
    """qkv [N, L, 3W] -> [N, L, W]; heads of 64."""

    @staticmethod
    def forward(ctx, qkv, heads, causal):
        _chk(qkv)
        qkv = qkv.contiguous()
        N, L, W3 = qkv.shape
        W = W3 // 3
        out = torch.empty(N, L, W, device=qkv.device, dtype=torch.float32)
        h2_mark_next(out)
        call("tris_mha_fwd_f32", P(qkv), P(out), N, L, W, heads, int(causal), _stream())
        ctx.cfg = (N, L, W, heads, int(causal))
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        N, L, W, heads, causal = ctx.cfg
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        h2_mark_next(dqkv)
        call("tris_mha_bwd_f32", P(qkv), P(do), P(dqkv), N, L, W, heads, causal, _stream())
        return dqkv, None, None


MHA_STATS = {"last": None}     # arithmetic of the last flash-style forward: "h2" | "f32" (tests)


class MhaMfmaFn(torch.autograd.Function):
    """Flash-style attention on the f32 MFMA, any L (csrc/attn_mfma.hip); saves the per-query log-sum-exp."""

    @staticmethod
    def forward(ctx, qkv, heads, causal):
        _chk(qkv)
        qkv = qkv.contiguous()
        N, L, W3 = qkv.shape
        W = W3 // 3
        out = torch.empty(N, L, W, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(N, heads, L, device=qkv.device, dtype=torch.float32)
        # inside an h2 step the products run on two fp16 pieces per operand (csrc/attn_h2.hip, 16-bit MFMA), scaled by qkv's amax word
        # (from L = 32 on: at the text encoder's L = 20 the split while staging costs more than the 16-bit MFMA saves, tools/mha_bench.py)
        w_qkv = _h2_amax(qkv) if (h2_on() and cfg.mha_h2 and L >= 32) else None
        h2_mark_next(out)
        if w_qkv is not None:
            _timed("mha_fwd", 4.0 * N * heads * L * L * 64 * (0.5 if causal else 1.0),
                   lambda: call("tris_mha_h2_fwd_f32", P(qkv), P(out), P(lse), w_qkv, N, L, W, heads, int(causal), _stream()))
        else:
            _timed("mha_fwd", 4.0 * N * heads * L * L * 64 * (0.5 if causal else 1.0),
                   lambda: call("tris_mha_mfma_fwd_f32", P(qkv), P(out), P(lse), N, L, W, heads, int(causal), _stream()))
        MHA_STATS["last"] = "h2" if w_qkv is not None else "f32"
        ctx.h2 = w_qkv is not None
        ctx.cfg = (N, L, W, heads, int(causal))
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, do):
        qkv, out, lse = ctx.saved_tensors
        N, L, W, heads, causal = ctx.cfg
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        w_qkv = w_do = None
        if ctx.h2 and h2_on():
            w_qkv, w_do = _h2_amax(qkv), _h2_amax(do)
        h2_mark_next(dqkv)
        if w_qkv is not None and w_do is not None:
            call("tris_mha_h2_bwd_f32", P(qkv), P(out), P(do), P(lse), P(delta), P(dqkv), w_qkv, w_do, N, L, W, heads, causal, _stream())
        else:
            call("tris_mha_mfma_bwd_f32", P(qkv), P(out), P(do), P(lse), P(delta), P(dqkv), N, L, W, heads, causal, _stream())
        return dqkv, None, None


def _mha_impl(L):
    """'valu' = the LDS-resident kernel for L <= 64 (csrc/attn.hip), 'mfma' = the flash-style MFMA kernel (any L).
    TRIS_MHA=valu|mfma forces one; default 'auto' takes the measured faster one: at L = 20 (text) the op is an HBM stream
    of the packed QKV and the single-launch backward of the LDS-resident kernel wins (654 vs 806 us at N = 3840), from
    L = 50 (aux ViT) on the MFMA kernel does (fwd 47 vs 53 us, bwd 126 vs 159 us; L = 401: 73 / 84 TFLOP/s)."""
    import os
    mode = cfg.mha
    if L > 64 or mode == "mfma":
        return "mfma"
    if mode == "valu":
        return "valu"
    return "valu" if L < 40 else "mfma"


def mha(qkv, heads, causal):
    if _mha_impl(qkv.shape[1]) == "valu":
        return MhaFn.apply(qkv, heads, causal)
    return MhaMfmaFn.apply(qkv, heads, causal)


class QGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ew("TRIS_EW_QGELU", x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ew("TRIS_EW_QGELU_BWD", dy.contiguous(), x)


def quick_gelu(x):
    return QGeluFn.apply(x)


class EmbedFn(torch.autograd.Function):
    """token_embedding(ids) + positional_embedding[:L]  (CLIP/clip/model.py:553-554).

    Backward: the positional gradient is a column sum; the token gradient is built from the ROW LIST (ids, d_out rows) by a
    deterministic kernel (tris_embed_rows_bwd_f32).  Data-parallel (`reducer` given and it takes the table): the list is handed
    to the reducer, which all-gathers the <= B*L rows of every rank and builds the SAME mean gradient on every rank -- 2 MB per
    rank on the wire instead of the dense 101 MB table."""

    @staticmethod
    def forward(ctx, ids, tok, pos, reducer=None):
        _chk(ids, tok, pos)
        ids = ids.contiguous()
        N, L = ids.shape
        W = tok.shape[1]
        out = torch.empty(N, L, W, device=tok.device, dtype=torch.float32)
        call("tris_embed_fwd_f32", P(ids), P(tok), P(pos), P(out), N, L, W, _stream())
        ctx.params = (tok, pos)
        ctx.reducer = reducer
        ctx.save_for_backward(ids, tok, pos)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, tok, pos = ctx.saved_tensors
        N, L = ids.shape
        W = tok.shape[1]
        dout = dout.contiguous()
        st = _sink(ctx.params[0]), _sink(ctx.params[1])
        dtok = st[0] if st[0] is not None else torch.empty_like(tok)
        dpos = st[1] if st[1] is not None else torch.empty_like(pos)
        dtok.zero_()
        dpos.zero_()
        call("tris_embed_bwd_f32", P(ids), P(dout), None, P(dpos), N, L, W, _stream())     # positional part
        red = ctx.reducer
        if red is not None and st[0] is not None and W % 4 == 0 and W <= 2048 and red.take_embedding_rows(ctx.params[0], ids.view(-1), dout.view(N * L, W), dtok):
            pass     # the reducer gathers every rank's rows and scatters them (scale 1 / world) into dtok
        elif W % 4 == 0 and W <= 2048:
            call("tris_embed_rows_bwd_f32", P(ids), P(dout), P(dtok), N * L, W, 1.0, _stream())
        else:
            call("tris_embed_bwd_f32", P(ids), P(dout), P(dtok), None, N, L, W, _stream())
        return None, (None if st[0] is not None else dtok), (None if st[1] is not None else dpos), None


def embed(ids, tok, pos, reducer=None):
    return EmbedFn.apply(ids, tok, pos, reducer)


class EotGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, x):
        _chk(ids, x)
        ids, x = (None if ids is None else ids.contiguous()), x.contiguous()
        N, L, W = x.shape
        out = torch.empty(N, W, device=x.device, dtype=torch.float32)
        call("tris_eot_gather_fwd_f32", P(ids), P(x), P(out), N, L, W, _stream())
        ctx.save_for_backward(ids)
        ctx.dims = (N, L, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        N, L, W = ctx.dims
        dx = torch.empty(N, L, W, device=dout.device, dtype=torch.float32)
        call("tris_eot_gather_bwd_f32", P(ids), P(dout.contiguous()), P(dx), N, L, W, _stream())
        return None, dx


def eot_gather(ids, x):
    return EotGatherFn.apply(ids, x)


# ---- packed text rows (csrc/attn.hip "packed text rows"; include/tris_hip.h) -- forward only, no autograd: the frozen aux tower
class rows_limit:
    """inside: this thread's tris_gemm_f32 (row-major A) and LayerNorm-forward launches skip rows >= plan[1] (a device word)"""

    def __init__(self, plan):
        self.ptr = plan.data_ptr() + 4

    def __enter__(self):
        call("tris_rows_limit_thread", self.ptr)

    def __exit__(self, *exc):
        call("tris_rows_limit_thread", None)
        return False


def packed_rows(N, L):
    """rows of every packed buffer: N L rounded up to the 256-row granule of the row limit"""
    return (N * L + 255) // 256 * 256


def text_packable(L):
    return not _BATCH_INVARIANT and 1 <= L <= 64


def text_pack_plan(ids):
    """ids int64 [N, L] -> plan (int32 device array: rows in use, row limit, first row of every sentence, source token of every row)"""
    N, L = ids.shape
    plan = torch.empty(3 + N + packed_rows(N, L), device=ids.device, dtype=torch.int32)
    call("tris_text_pack_plan_i64", P(ids), N, L, plan.data_ptr(), _stream())
    return plan


def embed_packed(ids, tok, pos, plan):
    N, L = ids.shape
    W = tok.shape[1]
    out = torch.empty(packed_rows(N, L), W, device=tok.device, dtype=torch.float32)
    call("tris_embed_packed_fwd_f32", P(ids), P(tok), P(pos), plan.data_ptr(), P(out), N, L, W, _stream())
    return out


def mha_packed(qkv, plan, N, L, heads, causal):
    """qkv [N L, 3 W] packed rows -> out [N L, W]; sentence n attends inside its own row range (L: the longest possible sentence)"""
    W = qkv.shape[-1] // 3
    out = torch.empty(qkv.shape[0], W, device=qkv.device, dtype=torch.float32)
    h2_mark_next(out)
    call("tris_mha_packed_fwd_f32", P(qkv), P(out), plan.data_ptr(), N, L, W, heads, int(causal), _stream())
    return out


def eot_gather_packed(x, plan, N):
    out = torch.empty(N, x.shape[-1], device=x.device, dtype=torch.float32)
    call("tris_eot_gather_packed_f32", P(x), plan.data_ptr(), P(out), N, x.shape[-1], _stream())
    return out


def token0(x):
    """x[:, 0, :] of [N, L, W] as a copy, with its own backward kernel (the ViT class token, CLIP/clip/model.py:443)"""
    return EotGatherFn.apply(None, x)


# ----------------------------------------------------------------------------------------------- heads
class L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = x.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        inv = torch.empty(rows, device=x.device, dtype=torch.float32)
        call("tris_l2norm_fwd_f32", P(x), P(y), P(inv), rows, C, _stream())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        call("tris_l2norm_bwd_f32", P(dy.contiguous()), P(y), P(inv), P(dx), inv.numel(), y.shape[-1], _stream())
        return dx


def l2norm(x):
    return L2NormFn.apply(x)


class SoftmaxFn(torch.autograd.Function):
    """softmax(scale * x) over the last dim"""

    @staticmethod
    def forward(ctx, x, scale):
        _chk(x)
        x = x.contiguous()
        n = x.shape[-1]
        y = torch.empty_like(x)
        call("tris_softmax_fwd_f32", P(x), P(y), x.numel() // n, n, scale, _stream())
        ctx.scale = scale
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        n = y.shape[-1]
        dx = torch.empty_like(y)
        call("tris_softmax_bwd_f32", P(dy.contiguous()), P(y), P(dx), y.numel() // n, n, ctx.scale, _stream())
        return dx, None


def softmax(x, scale=1.0):
    return SoftmaxFn.apply(x, scale)


class InstNormFn(torch.autograd.Function):
    """InstanceNorm2d(affine) [+ReLU] on channels-last x [B, P, C]"""

    @staticmethod
    def forward(ctx, x, g, b, relu, eps):
        _chk(x, g, b)
        x = x.contiguous()
        B, Pp, C = x.shape
        y = torch.empty_like(x)
        st = torch.empty(2, B, C, device=x.device, dtype=torch.float32)
        h2_mark_next(y)     # (h2: the launch leaves y's amax -- its consumers are dense products)
        call("tris_instnorm_fwd_f32", P(x), P(g), P(b), P(y), P(st), P(st, B * C), B, Pp, C, eps, int(relu), _stream())
        ctx.relu = bool(relu)
        ctx.params = (g, b)
        ctx.save_for_backward(x, g, b, st, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, st, y = ctx.saved_tensors
        B, Pp, C = x.shape
        dx = torch.empty_like(x)
        parts = torch.empty(2, B, C, device=x.device, dtype=torch.float32)
        h2_mark_next(dx)
        call("tris_instnorm_bwd_f32", P(dy.contiguous()), P(y), P(x), P(g), P(st), P(st, B * C), P(dx), P(parts),
             P(parts, B * C), B, Pp, C, int(ctx.relu), _stream())
        dg = _emit(ctx.params[0], lambda o: colsum(parts[0], B, C, o), ctx.needs_input_grad[1], reads=(parts,))
        db = _emit(ctx.params[1], lambda o: colsum(parts[1], B, C, o), ctx.needs_input_grad[2], reads=(parts,))
        return dx, dg, db, None, None


def instance_norm(x, g, b, relu=False, eps=1e-5):
    return InstNormFn.apply(x, g, b, relu, eps)


_XATTN_SYNC = {}
_XATTN_SPARE = {}      # device -> zeroed buffers allocated OUTSIDE any stream capture, for streams first seen under one
_XATTN_WORDS = 16 + 8 * 64


def _xattn_sync(dev, B):
    """the fused cross-attention kernel's device-side bookkeeping (launch epoch, finish ticket, time-out flag, publish flags):
    zeroed once, then owned by the kernel.  One buffer per (device, stream): launches on one stream are ordered.
    Nothing is allocated -- or cleared -- inside a stream capture: a capture runs on a stream of its own, so its first launch meets a
    stream without a buffer; a buffer created there would live in the graph's private pool and its zero-fill would be a graph node,
    i.e. every replay would clear the STICKY time-out word (sync[2]) before anyone looked at it.  Eager launches (the priming steps
    every capture is preceded by) keep spare buffers ready; a capture takes one of those."""
    key = (dev, torch.cuda.current_stream().cuda_stream)
    n = max(int(query("tris_xattn_fused_sync_words", B)), int(query("tris_xattn_px_sync_words", B)), _XATTN_WORDS)
    capturing = torch.cuda.is_current_stream_capturing()
    spare = _XATTN_SPARE.setdefault(dev, [])
    if not capturing:
        while len(spare) < 3 or any(b.numel() < n for b in spare):
            spare[:] = [b for b in spare if b.numel() >= n]
            spare.append(torch.zeros(n, device=dev, dtype=torch.int32))
    t = _XATTN_SYNC.get(key)
    if t is None or t.numel() < n:
        if capturing:
            fit = [b for b in spare if b.numel() >= n]
            if not fit:
                raise RuntimeError("fused cross attention: no sync buffer for a stream first seen under capture "
                                   "(run one eager step with the largest batch before capturing)")
            t = fit[0]
            spare.remove(t)
        else:
            t = torch.zeros(n, device=dev, dtype=torch.int32)
        _XATTN_SYNC[key] = t
    return t


def xattn_timed_out(collective=False):
    """True if a wait inside any fused cross-attention launch gave up (host sync).  The word is sticky: nothing but this process's
    exit clears it.  collective=True (every rank calls it at the same point): the answer is the group's, so that all ranks stop
    together instead of one raising while its peers wait for it in a collective."""
    hit = any(int(t[2].item()) != 0 for t in _XATTN_SYNC.values())
    if collective and torch.distributed.is_available() and torch.distributed.is_initialized():
        f = torch.tensor([1.0 if hit else 0.0], device="cuda" if torch.cuda.is_available() else "cpu")
        torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MAX)
        hit = bool(f.item() > 0)
    return hit


class XAttnFn(torch.autograd.Function):
    """Fused bilateral cross attention (model/attn.py:117-128): forward = tris_xattn_fwd (two launches for the whole
    batch); backward = nine large single GEMMs on the saved probabilities + two softmax-backward kernels."""

    @staticmethod
    def forward(ctx, Qv, Kv, Vv, Qt, Kt, Vt):
        _chk(Qv, Kv, Vv, Qt, Kt, Vt)
        Qv, Kv, Vv, Qt, Kt, Vt = (t.contiguous() for t in (Qv, Kv, Vv, Qt, Kt, Vt))
        B, Pp, C = Qv.shape
        N = Qt.shape[0]
        dev = Qv.device
        new_vis = torch.empty(B, Pp, C, device=dev, dtype=torch.float32)
        new_lan = torch.empty(B, N, C, device=dev, dtype=torch.float32)
        probs = torch.empty(B, 4, Pp, N, device=dev, dtype=torch.float32)
        done = px_ran = False
        # ONE persistent launch where a kernel's domain covers the shape and all its workgroups are co-resident: cut by pixel rows
        # (csrc/xattn_px.hip, S <= 8 workgroups per image, one hand-off) or, failing that, by channel slices (csrc/xattn_fused.hip,
        # eight per image, three hand-offs); TRIS_XATTN_FUSED=0 forces the two-launch pair
        for kind, on in (("px", cfg.xattn_fused and cfg.xattn_px), ("fused", cfg.xattn_fused)):
            ws_bytes = query(f"tris_xattn_{kind}_ws_bytes", B, N, C) if on and not done else 0
            if ws_bytes > 0:
                ws = torch.empty(ws_bytes // 4 + 4, device=dev, dtype=torch.float32)
                sync = _xattn_sync(dev, B)
                if kind == "px" and h2_on() and cfg.xattn_h2:
                    # h2: the products of the launch run on two fp16 pieces per operand, scaled by the amax words the producing
                    # projections left behind (a pre-pass where there is none)
                    words = [_h2_amax(t) for t in (Qv, Kv, Vv, Qt, Kt, Vt)]
                    if all(w is not None for w in words):
                        call("tris_xattn_amax_next", *words)
                done = _timed(f"xattn_fwd_{kind}", 8.0 * B * Pp * N * C, lambda: _declinable(
                    f"tris_xattn_{kind}_fwd_f32", P(Qv), P(Kv), P(Vv), P(Qt), P(Kt), P(Vt), P(new_vis), P(new_lan), P(probs), B,
                    Pp, N, C, P(ws), ws.numel() * 4, sync.data_ptr(), _stream()))
                px_ran = done and kind == "px"
        if not done:
            _timed("xattn_fwd_pair", 8.0 * B * Pp * N * C,
                   lambda: call("tris_xattn_fwd_f32", P(Qv), P(Kv), P(Vv), P(Qt), P(Kt), P(Vt), P(new_vis), P(new_lan),
                                P(probs), B, Pp, N, C, _stream()))
        ctx.dims = (B, Pp, N, C)
        ctx.px_form = px_ran
        ctx.save_for_backward(Qv, Kv, Vv, Qt, Kt, Vt, probs)
        return new_vis, new_lan

    @staticmethod
    def _backward_px(ctx, d_vis, d_lan):
        """the pixel-row form of the backward (csrc/xattn_px.hip): one persistent launch for dQv, dKv, dVv and the two soft-max
        backwards, then the three [N, C] sums over images and pixels as split-K products.  None: outside its domain."""
        Qv, Kv, Vv, Qt, Kt, Vt, probs = ctx.saved_tensors
        B, Pp, N, C = ctx.dims
        dev = Qv.device
        ws_bytes = query("tris_xattn_px_bwd_ws_bytes", B, N, C)
        if ws_bytes <= 0:
            return None
        ws = torch.empty(ws_bytes // 4 + 4, device=dev, dtype=torch.float32)
        sync = _xattn_sync(dev, B)
        dQv, dKv, dVv = (torch.empty(B, Pp, C, device=dev, dtype=torch.float32) for _ in range(3))
        dS = torch.empty(3, B * Pp, N, device=dev, dtype=torch.float32)
        if h2_on() and cfg.xattn_h2:
            words = [_h2_amax(t) for t in (d_vis, Vv, d_lan, Qt, Kt, Vt)]
            if all(w is not None for w in words):
                call("tris_xattn_amax_next", *words)
        slots = [_h2_slot() if h2_on() else None for _ in range(3)]   # the launch leaves the amax words of its three outputs
        if not _timed("xattn_bwd_px", 10.0 * B * Pp * N * C, lambda: _declinable(
                "tris_xattn_px_bwd_f32", P(d_vis), P(d_lan), P(Vv), P(Qt), P(Kt), P(Vt), P(probs), P(dQv), P(dKv), P(dVv), P(dS), B, Pp,
                N, C, P(ws), ws.numel() * 4, sync.data_ptr(), slots[0], slots[1], slots[2], _stream())):
            return None
        for t, slot in zip((dQv, dKv, dVv), slots):
            if slot is not None:
                t._h2 = (_H2["step"], slot, t._version)
        BP = B * Pp
        new = lambda: torch.empty(N, C, device=dev, dtype=torch.float32)
        dVt = gemm(dS[2], d_vis, new(), N, C, BP, N, C, C, True, False)       # Av^T . d_vis
        dKt = gemm(dS[0], Qv, new(), N, C, BP, N, C, C, True, False)          # dS1^T . Qv
        dQt = gemm(dS[1], Kv, new(), N, C, BP, N, C, C, True, False)          # dS2^T . Kv
        return dQv, dKv, dVv, dQt, dKt, dVt

    @staticmethod
    def backward(ctx, d_vis, d_lan):
        Qv, Kv, Vv, Qt, Kt, Vt, probs = ctx.saved_tensors
        B, Pp, N, C = ctx.dims
        dev = Qv.device
        scale = 1.0 / math.sqrt(C)
        d_vis, d_lan = d_vis.contiguous(), d_lan.contiguous()
        if ctx.px_form and cfg.xattn_bwd_px:
            got = XAttnFn._backward_px(ctx, d_vis, d_lan)
            if got is not None:
                return got
        BP = B * Pp
        Av = probs[:, 0].contiguous().view(BP, N)
        AtT = probs[:, 2].contiguous().view(B, Pp, N)

        def new(*shape):
            return torch.empty(*shape, device=dev, dtype=torch.float32)
        # pixel -> sentence direction:  new_vis = Av . Vt
        dAv = gemm(d_vis, Vt, new(BP, N), BP, N, C, C, C, N, False, True)                      # [BP,N] = d_vis . Vt^T
        dVt = gemm(Av, d_vis, new(N, C), N, C, BP, N, C, C, True, False)                       # Av^T . d_vis
        dS1 = new(BP, N)
        call("tris_softmax_bwd_f32", P(dAv), P(Av), P(dS1), BP, N, scale, _stream())
        dQv = gemm(dS1, Kt, new(B, Pp, C), BP, C, N, N, C, C, False, False)                    # dS1 . Kt
        dKt = gemm(dS1, Qv, new(N, C), N, C, BP, N, C, C, True, False)                         # dS1^T . Qv
        # sentence -> pixel direction:  new_lan[b] = AtT[b]^T . Vv[b]
        dAtT = gemm(Vv, d_lan, new(B, Pp, N), Pp, N, C, C, C, N, False, True, batch=B, sA=Pp * C, sB=N * C, sC=Pp * N)
        dVv = gemm(AtT, d_lan, new(B, Pp, C), Pp, C, N, N, C, C, False, False, batch=B, sA=Pp * N, sB=N * C, sC=Pp * C)
        dS2 = new(B, Pp, N)
        call("tris_softmax_col_bwd_f32", P(dAtT), P(AtT), P(dS2), B, Pp, N, scale, _stream())
        dKv = gemm(dS2, Qt, new(B, Pp, C), BP, C, N, N, C, C, False, False)                    # dS2T . Qt
        dQt = gemm(dS2, Kv, new(N, C), N, C, BP, N, C, C, True, False)                         # dS2T^T . Kv
        return dQv, dKv, dVv, dQt, dKt, dVt


def xattn(Qv, Kv, Vv, Qt, Kt, Vt):
    return XAttnFn.apply(Qv, Kv, Vv, Qt, Kt, Vt)


class AxpyFn(torch.autograd.Function):
    """s * a + b.  grad_box_b: b's gradient (dy itself) is left in that GradBox for another consumer of b to add in its
    data-gradient epilogue instead of being summed by autograd."""

    @staticmethod
    def forward(ctx, a, b, s, grad_box_b=None):
        ctx.s, ctx.box = s, grad_box_b
        return ew("TRIS_EW_AXPY", a.contiguous(), b.contiguous(), s)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        da = ew("TRIS_EW_SCALE", dy, None, ctx.s) if ctx.needs_input_grad[0] else None
        db = dy if ctx.needs_input_grad[1] else None
        if db is not None and ctx.box is not None and ctx.box.deposit(db):
            db = None
        return da, db, None, None


def axpy(a, b, s, grad_box_b=None):
    return AxpyFn.apply(a, b, s, grad_box_b)


class AxpyBcastFn(torch.autograd.Function):
    """s * a + b with b [n...] broadcast over the leading dimension of a [R, n...]  (model_stage1.py:74: the sentence features of
    the step against every image's attended ones; the expanded copy of b is never made)"""

    @staticmethod
    def forward(ctx, a, b, s, grad_box_b=None):
        _chk(a, b)
        a, b = a.contiguous(), b.contiguous()
        assert a.shape[1:] == b.shape and b.numel() % 4 == 0, (a.shape, b.shape)
        ctx.s, ctx.box = s, grad_box_b
        out = torch.empty_like(a)
        h2_mark_next(out)
        call("tris_elementwise_bcast_f32", EW["TRIS_EW_AXPY"], P(a), P(b), P(out), a.numel(), b.numel(), float(s), _stream())
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        da = ew("TRIS_EW_SCALE", dy, None, ctx.s) if ctx.needs_input_grad[0] else None
        db = None
        if ctx.needs_input_grad[1]:
            R = dy.shape[0]
            db = colsum(dy, R, dy.numel() // R, torch.empty(dy.shape[1:], device=dy.device, dtype=torch.float32))
            if ctx.box is not None and ctx.box.deposit(db):
                db = None
        return da, db, None, None


def axpy_bcast(a, b, s, grad_box_b=None):
    return AxpyBcastFn.apply(a, b, s, grad_box_b)


class ScaleExpFn(torch.autograd.Function):
    """(x, ls) -> (x * exp(ls), exp(ls)) for a 0-dim device scalar ls  (model_stage1.py:77-78)"""

    @staticmethod
    def forward(ctx, x, ls):
        _chk(x, ls)
        x = x.contiguous()
        out = torch.empty_like(x)
        e = torch.empty((), device=x.device, dtype=torch.float32)
        call("tris_scale_exp_fwd_f32", P(x), P(ls), P(out), P(e), x.numel(), _stream())
        ctx.save_for_backward(out, ls, e)
        ctx.set_materialize_grads(False)
        return out, e

    @staticmethod
    def backward(ctx, dout, de):
        out, ls, e = ctx.saved_tensors
        dx = dls = None
        if dout is not None:
            dout = dout.contiguous()
            dx = torch.empty_like(dout) if ctx.needs_input_grad[0] else None
            dls = torch.empty((), device=out.device, dtype=torch.float32)
            ws = workspace(query("tris_scale_exp_workspace_bytes"))
            call("tris_scale_exp_bwd_f32", P(dout), P(out), P(ls), P(dx), P(dls), P(ws), out.numel(), _stream())
        if de is not None:       # (nobody differentiates through the returned scale on this path; kept correct all the same)
            dls = de * e if dls is None else dls + de * e
        return dx, (dls if ctx.needs_input_grad[1] else None)


def scale_exp(x, ls):
    return ScaleExpFn.apply(x, ls)


def concat_i64(a, b):
    """[a; b] of two int64 row lists (no gradient)"""
    a, b = a.contiguous(), b.contiguous()
    assert a.dtype == b.dtype == torch.int64 and a.shape[1:] == b.shape[1:] and a.is_cuda and b.is_cuda
    out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), device=a.device, dtype=torch.int64)
    call("tris_concat_i64", a.data_ptr(), a.numel(), b.data_ptr(), b.numel(), out.data_ptr(), _stream())
    return out


class MulFn(torch.autograd.Function):
    """a * b (same shape)"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ew("TRIS_EW_MUL", a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        return (ew("TRIS_EW_MUL", dy, b) if ctx.needs_input_grad[0] else None,
                ew("TRIS_EW_MUL", dy, a) if ctx.needs_input_grad[1] else None)


def mul(a, b):
    return MulFn.apply(a, b)


class ScoreHeadsFn(torch.autograd.Function):
    """score [B, P, N] -> (cls_out [B,N], cls_fg [B], relu_map [B,1,S,S], sig_map [B,1,S,S]) for training,
    or only relu_map in eval.  model_stage1.py:80-119."""

    @staticmethod
    def forward(ctx, score, h, w, S, train, focal_p, focal_c):
        _chk(score)
        score = score.contiguous()
        B, Pp, N = score.shape
        dev = score.device
        relu_map = torch.empty(B, 1, S, S, device=dev, dtype=torch.float32)
        ctx.cfg = (B, Pp, N, h, w, S, train, focal_p, focal_c)
        ctx.save_for_backward(score)
        if not train:
            call("tris_maps_fwd_f32", P(score), P(relu_map), None, B, h, w, N, S, _stream())
            return relu_map
        cls_out = torch.empty(B, N, device=dev, dtype=torch.float32)
        cls_fg = torch.zeros(B, device=dev, dtype=torch.float32)
        sig_map = torch.empty(B, 1, S, S, device=dev, dtype=torch.float32)
        call("tris_cls_head_fwd_f32", P(score), P(cls_out), P(cls_fg), B, Pp, N, focal_p, focal_c, _stream())
        call("tris_maps_fwd_f32", P(score), P(relu_map), P(sig_map), B, h, w, N, S, _stream())
        ctx.mark_non_differentiable(cls_fg)
        return cls_out, cls_fg, relu_map, sig_map

    @staticmethod
    def backward(ctx, *grads):
        (score,) = ctx.saved_tensors
        B, Pp, N, h, w, S, train, focal_p, focal_c = ctx.cfg
        if train:
            g_cls, _, g_relu, g_sig = grads
        else:
            g_cls, g_relu, g_sig = None, grads[0], None
        dscore = torch.empty_like(score)
        if g_cls is not None:
            call("tris_cls_head_bwd_f32", P(score), P(g_cls.contiguous()), P(dscore), B, Pp, N, focal_p, focal_c,
                 _stream())
        else:
            dscore.zero_()
        if g_relu is not None or g_sig is not None:
            call("tris_maps_bwd_f32", P(score), P(g_relu.contiguous()) if g_relu is not None else None,
                 P(g_sig.contiguous()) if g_sig is not None else None, P(dscore), B, h, w, N, S, _stream())
        return dscore, None, None, None, None, None, None


def score_heads(score, h, w, S, train, focal_p=3.0, focal_c=0.01):
    return ScoreHeadsFn.apply(score, h, w, S, train, focal_p, focal_c)


class ResizeFn(torch.autograd.Function):
    """F.interpolate(x [B,C,H,W], size, mode='bilinear', align_corners=align)"""

    @staticmethod
    def forward(ctx, x, Ho, Wo, align):
        _chk(x)
        x = x.contiguous()
        B, C, Hi, Wi = x.shape
        y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
        call("tris_resize_bilinear_fwd_f32", P(x), P(y), B * C, Hi, Wi, Ho, Wo, int(align), _stream())
        ctx.cfg = (B, C, Hi, Wi, Ho, Wo, int(align))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, Hi, Wi, Ho, Wo, align = ctx.cfg
        dx = torch.empty(B, C, Hi, Wi, device=dy.device, dtype=torch.float32)
        call("tris_resize_bilinear_bwd_f32", P(dy.contiguous()), P(dx), B * C, Hi, Wi, Ho, Wo, align, _stream())
        return dx, None, None, None


def resize_bilinear(x, size, align_corners):
    return ResizeFn.apply(x, int(size[0]), int(size[1]), bool(align_corners))


class FgPatchFn(torch.autograd.Function):
    """(cam [B,1,R,R], img [B,C,R,R]) -> fg = cam*img as ViT patch-GEMM rows [B, (R/ps)^2, C*ps*ps]"""

    @staticmethod
    def forward(ctx, cam, img, ps):
        _chk(cam, img)
        cam, img = cam.contiguous(), img.contiguous()
        B, C, R, _ = img.shape
        G = R // ps
        out = torch.empty(B, G * G, C * ps * ps, device=img.device, dtype=torch.float32)
        call("tris_fg_patch_fwd_f32", P(cam), P(img), P(out), B, C, R, ps, _stream())
        ctx.cfg = (B, C, R, ps)
        ctx.save_for_backward(img)
        return out

    @staticmethod
    def backward(ctx, dp):
        (img,) = ctx.saved_tensors
        B, C, R, ps = ctx.cfg
        dcam = torch.empty(B, 1, R, R, device=dp.device, dtype=torch.float32)
        call("tris_fg_patch_bwd_f32", P(dp.contiguous()), P(img), P(dcam), B, C, R, ps, _stream())
        return dcam, None, None


def fg_patches(cam, img, ps):
    return FgPatchFn.apply(cam, img, ps)


class VitAssembleFn(torch.autograd.Function):
    """[cls + pos[0]; emb + pos[1:]] -> [B, T, W]; gradients for cls / pos only where they are trained (ViT trunk)"""

    @staticmethod
    def forward(ctx, emb, cls, pos):
        _chk(emb, cls, pos)
        emb = emb.contiguous()
        B, T1, W = emb.shape
        x = torch.empty(B, T1 + 1, W, device=emb.device, dtype=torch.float32)
        call("tris_vit_assemble_fwd_f32", P(emb), P(cls), P(pos), P(x), B, T1 + 1, W, _stream())
        ctx.cfg = (B, T1 + 1, W)
        return x

    @staticmethod
    def backward(ctx, dx):
        B, T, W = ctx.cfg
        dx = dx.contiguous()
        demb = torch.empty(B, T - 1, W, device=dx.device, dtype=torch.float32)
        call("tris_vit_assemble_bwd_f32", P(dx), P(demb), B, T, W, _stream())
        dcls = dpos = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dpos = colsum(dx, B, T * W, torch.empty(T, W, device=dx.device, dtype=torch.float32))   # sum over the batch
            dcls = dpos[0].clone() if ctx.needs_input_grad[1] else None
            dpos = dpos if ctx.needs_input_grad[2] else None
        return demb, dcls, dpos


def vit_assemble(emb, cls, pos):
    return VitAssembleFn.apply(emb, cls, pos)


class Stage1LossFn(torch.autograd.Function):
    """(cls [B,N], f_img [B,E], f_txt [B,E], f_neg [B,K,E]|None) -> losses[4] = (total, l1, l4, l5).
    train_stage1.py:263-284, 340-364.  Text features come from the frozen aux CLIP: no gradient to them."""

    @staticmethod
    def forward(ctx, cls, fi, ft, fneg, w1, w4, w5):
        _chk(cls, fi, ft, fneg)
        cls, fi, ft = cls.contiguous(), fi.contiguous(), ft.contiguous()
        B, N = cls.shape
        E = fi.shape[1]
        K = 0
        if fneg is not None:
            fneg = fneg.contiguous()
            K = fneg.shape[1]
        dev = cls.device
        scratch = torch.empty(B * 4, device=dev, dtype=torch.float32)
        losses = torch.empty(4, device=dev, dtype=torch.float32)
        call("tris_stage1_loss_fwd_f32", P(cls), P(fi), P(ft), P(fneg), B, N, E, K, w1, w4, w5, P(scratch),
             P(scratch, 3 * B), P(losses), _stream())
        ctx.cfg = (B, N, E, K, w1, w4, w5)
        ctx.save_for_backward(cls, fi, ft, fneg)
        return losses

    @staticmethod
    def backward(ctx, g):
        cls, fi, ft, fneg = ctx.saved_tensors
        B, N, E, K, w1, w4, w5 = ctx.cfg
        g = g.contiguous()
        # (the kernels fold the loss weights: dL/dl1 = g[0]*w1 + g[1]; dL/dl5 = g[0]*w5 + g[3]; dL/dl4 = g[0]*w4 + g[2])
        dcls = torch.empty_like(cls)
        dfi = torch.empty_like(fi)
        call("tris_stage1_loss_bwd_f32", P(cls), P(fi), P(ft), P(fneg), P(g), float(w1), float(w4), float(w5), B, N, E, K, P(dcls),
             P(dfi), _stream())
        return dcls, dfi, None, None, None, None, None


def stage1_loss(cls, fi, ft, fneg, w1=1.0, w4=5.0, w5=2.0):
    return Stage1LossFn.apply(cls, fi, ft, fneg, w1, w4, w5)


def eval_post(relu_map, target_u8):
    """validate.py:180-190 for one (image, sentence): returns (iu int64[3] on device = I, U, argmax; cam [oH,oW])."""
    _chk(relu_map, target_u8)
    S = relu_map.shape[-1]
    oH, oW = target_u8.shape[-2:]
    cam = torch.empty(oH, oW, device=relu_map.device, dtype=torch.float32)
    iu = torch.empty(3, device=relu_map.device, dtype=torch.int64)
    ws = workspace(4 * (2 * 1024 + 8))
    call("tris_eval_post_f32", P(relu_map.contiguous()), S, P(target_u8.contiguous()), oH, oW, P(cam), P(iu), P(ws),
         _stream())
    return iu, cam


# ------------------------------------------------------------------------------------------------------ input pipeline
#  uint8 dataset resident in HBM; see tris_amd/dataset/hbm.py
_PIL_TABLES = {}


def _pil_tables(kind, n_in, n_out, device):
    key = (kind, n_in, n_out, str(device))
    t = _PIL_TABLES.get(key)
    if t is None:
        from .dataset import pil_tables
        if kind == "bilinear":
            bounds, kk, ksize = pil_tables.resample_tables(n_in, n_out)
            t = (torch.from_numpy(bounds.copy()).to(device), torch.from_numpy(kk.copy()).to(device), ksize)
        else:
            t = (torch.from_numpy(pil_tables.nearest_index(n_in, n_out).copy()).to(device),)
        _PIL_TABLES[key] = t
    return t


def _chk_u8(*ts):
    for t in ts:
        if not t.is_cuda:
            raise NoGpuError("tris_amd ops run on MI355X only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.uint8:
            raise TypeError(f"expected uint8, got {t.dtype}")


def resample_u8(img, out_h, out_w, out=None):
    """Pillow `resize((out_w, out_h), BILINEAR)` of a uint8 [H,W,C] device image, bit-exact (dataset/transform.py:29)."""
    _chk_u8(img)
    img = img.contiguous()
    H, W, C = img.shape
    if out is None:
        out = torch.empty(out_h, out_w, C, dtype=torch.uint8, device=img.device)
    if (H, W) == (out_h, out_w):
        out.copy_(img)
        return out
    th = _pil_tables("bilinear", W, out_w, img.device) if W != out_w else (None, None, 0)
    tv = _pil_tables("bilinear", H, out_h, img.device) if H != out_h else (None, None, 0)
    tmp = torch.empty(H * out_w * C, dtype=torch.uint8, device=img.device) if (th[0] is not None and tv[0] is not None) \
        else None
    call("tris_resample_u8", P(img), H, W, C, P(th[0]), P(th[1]), th[2], P(tv[0]), P(tv[1]), tv[2], out_h, out_w,
         P(tmp), P(out), _stream())
    return out


def resize_nearest_u8(img, out_h, out_w, out=None):
    """Pillow `resize((out_w, out_h), NEAREST)` of a uint8 [H,W] / [H,W,C] device image (dataset/transform.py:32)."""
    _chk_u8(img)
    img = img.contiguous()
    H, W = img.shape[:2]
    C = img.shape[2] if img.dim() == 3 else 1
    if out is None:
        out = torch.empty((out_h, out_w) + tuple(img.shape[2:]), dtype=torch.uint8, device=img.device)
    if (H, W) == (out_h, out_w):
        out.copy_(img)
        return out
    yi, = _pil_tables("nearest", H, out_h, img.device)
    xi, = _pil_tables("nearest", W, out_w, img.device)
    call("tris_gather2d_u8", P(img), H, W, C, P(yi), P(xi), out_h, out_w, P(out), _stream())
    return out


def gather_normalize(cache, index, lut, planar=False):
    """cache uint8 [N,H,W,3], index int64 [B], lut float32 [3,256] -> float32 batch of ToTensor+Normalize'd images:
    planar=False: memory [B,H,W,3], returned as the [B,3,H,W] channels-last view;  planar=True: contiguous [B,3,H,W]."""
    _chk_u8(cache)
    _chk(index, lut)
    N, H, W, C = cache.shape
    assert C == 3 and cache.is_contiguous() and index.dtype == torch.int64
    B = index.numel()
    if planar:
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=cache.device)
    else:
        out = torch.empty(B, H, W, 3, dtype=torch.float32, device=cache.device)
    call("tris_u8_gather_normalize_f32", P(cache), P(index), B, H * W, P(lut.contiguous()), P(out), int(planar), _stream())
    return out if planar else out.permute(0, 3, 1, 2)


def gather_rows(table, index):
    """table [N, ...] (any dtype, row size a multiple of 4 bytes), index int64 [R] -> [R, ...]"""
    if not table.is_cuda:
        raise NoGpuError("tris_amd ops run on MI355X only (got a CPU tensor); there is no CPU fallback")
    assert table.is_contiguous() and index.dtype == torch.int64 and index.is_cuda
    row_bytes = table[0].numel() * table.element_size()
    out = torch.empty((index.numel(),) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
    if index.numel():
        call("tris_gather_rows", P(table), P(index), index.numel(), row_bytes, P(out), _stream())
    return out
