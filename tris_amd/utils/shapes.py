"""State-dict specifications (key, shape, is_buffer) of the Stage-1 model and the aux CLIP, taken from the
tris_amd modules themselves (constructible on CPU); used by tests and the CPU-baseline leg of bench.py."""
import warnings

import torch


def _build_tris(extra=()):
    from ..args import get_parser
    from ..model.model_stage1 import TRIS
    args = get_parser().parse_args(["--backbone", "clip-RN50", "--size", "320", "--max_query_len", "20",
                                    "--negative_samples", "3"] + list(extra))
    from ..CLIP import clip
    with warnings.catch_warnings(), clip.random_init():   # architecture only: callers fill the weights (seed_fill)
        warnings.simplefilter("ignore")
        return TRIS(args)


def _spec(m):
    bufs = {k for k, _ in m.named_buffers()}
    return [(k, tuple(v.shape), k in bufs) for k, v in m.state_dict().items()]


def tris_state_dict_spec():
    return _spec(_build_tris())


def aux_state_dict_spec(txt_length=20):
    from ..CLIP.clip.model import ARCH, CLIP
    return _spec(CLIP(txt_length=txt_length, **ARCH["ViT-B/32"]))


def empty_state_dict(spec):
    """Plain contiguous CPU tensors for every key (int64 for num_batches_tracked)."""
    out = {}
    for k, shape, _ in spec:
        dt = torch.long if k.endswith("num_batches_tracked") else torch.float32
        out[k] = torch.zeros(shape, dtype=dt)
    out_ls = [k for k in out if k.endswith("logit_scale")]
    for k in out_ls:
        out[k] = torch.ones(()) * 2.6592600369327779  # log(1/0.07), the constructor value seed_fill keeps
    return out
