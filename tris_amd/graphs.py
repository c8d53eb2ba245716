"""hipGraph capture of the Stage-1 inference path (validate.py hot loop at batch 1 is launch-latency bound:
~450 short kernels per forward).  Two graphs, matching the loop structure of `validate`:

    visual(img)      RN50 trunk -> vis_project -> L2 norm        replayed once per image
    sentence(ids)    text encoder -> cross attention -> maps     replayed once per sentence of that image

Inputs are copied into static buffers; outputs are static buffers owned by the graphs (consume or clone them before
the next replay).  Capture uses torch's stream-capture plumbing (torch.cuda.graph); every captured node is one of
this repo's HIP kernels launched through the C ABI on the capturing stream."""
import torch


class GraphedStage1Eval:
    def __init__(self, net, img_shape, query_len, warmup=2):
        assert not net.training, "capture the eval path (model.eval())"
        self.net = net
        dev = next(net.parameters()).device
        self.s_img = torch.zeros(img_shape, device=dev, dtype=torch.float32)
        self.s_ids = torch.zeros(img_shape[0], query_len, device=dev, dtype=torch.int64)
        self.s_ids[:, 0] = 49406
        self.s_ids[:, 1] = 49407
        H = img_shape[2]
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):  # first-call setup (hipFuncSetAttribute, workspace growth) outside capture
                    v = net.encode_visual(self.s_img)
                    net.forward_cached(v, self.s_ids, H)
            torch.cuda.current_stream().wait_stream(side)
            self.g_vis = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_vis):
                self.vis = net.encode_visual(self.s_img)
            self.g_txt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_txt):
                self.out = net.forward_cached(self.vis, self.s_ids, H)

    def visual(self, img):
        self.s_img.copy_(img, non_blocking=True)
        self.g_vis.replay()

    def sentence(self, ids):
        self.s_ids.copy_(ids, non_blocking=True)
        self.g_txt.replay()
        return self.out


class GraphedFrozenText:
    """hipGraph replay of a FROZEN text tower (the aux CLIP of the Stage-1 loss, train_stage1.py:167, 346-347): ~130 short
    kernels per step whose only cost is the host time to issue them -- time during which the compute stream has nothing
    queued.  Static shapes ([n sentences, L tokens]), no gradient: captured once (after two eager warm-up calls, so that the
    GEMM autotuner has seen every shape), replayed as ONE launch per step on the caller's side stream.  The output is a static
    buffer: consume it before the next replay (the step's own stream order guarantees that)."""

    def __init__(self, clip_model, n, L, warmup=2):
        dev = next(clip_model.parameters()).device
        self.key = (n, L, clip_model.token_embedding.weight.data_ptr())
        self.s_ids = torch.zeros(n, L, device=dev, dtype=torch.int64)
        self.s_ids[:, 0] = 49406
        self.s_ids[:, 1] = 49407
        with torch.no_grad():
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                for _ in range(warmup):
                    clip_model.encode_text(self.s_ids)
            torch.cuda.current_stream().wait_stream(cap)
            self.g = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (the collective backend's watchdog, the autograd engine of a previous
            # step) may touch the device while this stream is capturing
            with torch.cuda.graph(self.g, capture_error_mode="thread_local"):
                self.out = clip_model.encode_text(self.s_ids)[1]

    def __call__(self, ids):
        """replay on the CURRENT stream; returns the static [n, E] output"""
        self.s_ids.copy_(ids, non_blocking=True)
        self.g.replay()
        return self.out


def frozen_text(clip_model, ids):
    """encode_text(ids)[1] of a frozen tower through a cached hipGraph (TRIS_HIPGRAPH=0: eager)"""
    import os
    if os.environ.get("TRIS_HIPGRAPH", "1") == "0" or torch.cuda.is_current_stream_capturing() or torch.is_grad_enabled():
        return clip_model.encode_text(ids)[1]
    n, L = ids.shape
    cache = clip_model.__dict__.setdefault("_tris_text_graphs", {})
    key = (n, L, clip_model.token_embedding.weight.data_ptr())
    g = cache.get(key)
    if g is None:
        if len(cache) >= 4:
            cache.clear()
        g = cache[key] = GraphedFrozenText(clip_model, n, L)
    return g(ids)
