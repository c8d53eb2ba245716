"""`clip.load` / `clip.tokenize` with the reference's signatures (CLIP/clip/clip.py:94-145, 200-240).

`load(name, device, jit=False, download_root=None, txt_length=77) -> (model, preprocess)`:
  * `name` is a checkpoint FILE (as train_stage1.py:167 relies on for "ViT-B-32") -> weights are loaded from it
    (TorchScript archive or plain state dict);
  * otherwise `name` is an architecture key ("RN50", "ViT-B/32", ...): the weights are looked up under
    `download_root` / ~/.cache/clip.  There is no network on the target machines, so nothing is downloaded; when the
    file is missing `load` RAISES (the reference would download or fail, clip.py:43-72) -- training or validating from
    a randomly initialised CLIP is never what a caller of the reference's scripts wants.  Tests, benchmarks and the
    synthetic smoke step opt in to the random-initialised architecture explicitly: TRIS_RANDOM_INIT=1 or
    `clip.allow_random_init(True)` (weights then come from `load_state_dict` / the seed-fill protocol).
The MI355X path keeps fp32 weights (the reference `.float()`s the Stage-1 model, model_stage1.py:31).
"""
import os
import warnings
from typing import List, Union

import torch

from ...config import cfg
from .model import ARCH, CLIP, build_model

__all__ = ["available_models", "load", "tokenize", "allow_random_init", "random_init"]
_tokenizer = None
_RANDOM_INIT_OK = None   # None: follow the environment (TRIS_RANDOM_INIT=1)


def allow_random_init(on=True):
    """explicit opt-in to building an architecture without its weights file (tests / benchmarks / smoke only)"""
    global _RANDOM_INIT_OK
    _RANDOM_INIT_OK = bool(on)


class random_init:
    """`with clip.random_init(): ...` -- scoped form of allow_random_init (restores the previous setting)"""

    def __enter__(self):
        global _RANDOM_INIT_OK
        self.prev, _RANDOM_INIT_OK = _RANDOM_INIT_OK, True

    def __exit__(self, *exc):
        global _RANDOM_INIT_OK
        _RANDOM_INIT_OK = self.prev


def _random_init_ok():
    return cfg.random_init if _RANDOM_INIT_OK is None else _RANDOM_INIT_OK


def _tok():
    global _tokenizer
    if _tokenizer is None:
        from .simple_tokenizer import SimpleTokenizer
        _tokenizer = SimpleTokenizer()
    return _tokenizer


def available_models() -> List[str]:
    return [k for k in ARCH if k != "ViT-B-32"]


def load(name, device="cuda" if torch.cuda.is_available() else "cpu", jit=False, download_root=None, txt_length=77):
    if os.path.isfile(name):
        try:
            sd = torch.jit.load(name, map_location="cpu").state_dict()
        except RuntimeError:
            sd = torch.load(name, map_location="cpu", weights_only=True)   # plain state dict: tensors only
            sd = sd.get("state_dict", sd)
        model = build_model(sd, txt_length=txt_length)
    elif name in ARCH:
        root = download_root or os.path.expanduser("~/.cache/clip")
        cand = os.path.join(root, name.replace("/", "-") + ".pt")
        if os.path.isfile(cand):
            return load(cand, device, jit, download_root, txt_length)
        if not _random_init_ok():
            raise FileNotFoundError(
                f"CLIP weights for {name} not found under {root} (looked for {cand}; there is no network to download "
                f"them).  Put the checkpoint there or pass its path as `name`.  Only tests / benchmarks build the "
                f"architecture without weights: set TRIS_RANDOM_INIT=1 or call clip.allow_random_init(True).")
        warnings.warn(f"CLIP weights for {name} not found under {root}: architecture built with random initialisation "
                      f"(explicitly allowed); call load_state_dict() to supply weights")
        model = CLIP(txt_length=txt_length, **ARCH[name]).eval()
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    return model.to(device).float(), None  # preprocess (torchvision transform) is not part of the hot path


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
    if isinstance(texts, str):
        texts = [texts]
    tk = _tok()
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    out = torch.zeros(len(texts), context_length, dtype=torch.int)
    for i, t in enumerate(texts):
        ids = [sot] + tk.encode(t) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, :len(ids)] = torch.tensor(ids)
    return out
