"""`clip.load` / `clip.tokenize` with the reference's signatures (CLIP/clip/clip.py:94-145, 200-240).

`load(name, device, jit=False, download_root=None, txt_length=77) -> (model, preprocess)`:
  * `name` is a checkpoint FILE (as train_stage1.py:167 relies on for "ViT-B-32") -> weights are loaded from it
    (TorchScript archive or plain state dict);
  * otherwise `name` is an architecture key ("RN50", "ViT-B/32", ...).  There is no network on the target
    machines, so instead of downloading, the architecture is built with its initialiser and a warning is
    emitted; load weights afterwards with `load_state_dict`.
The MI355X path keeps fp32 weights (the reference `.float()`s the Stage-1 model, model_stage1.py:31).
"""
import os
import warnings
from typing import List, Union

import torch

from .model import ARCH, CLIP, build_model

__all__ = ["available_models", "load", "tokenize"]
_tokenizer = None


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
            sd = torch.load(name, map_location="cpu")
            sd = sd.get("state_dict", sd)
        model = build_model(sd, txt_length=txt_length)
    elif name in ARCH:
        root = download_root or os.path.expanduser("~/.cache/clip")
        cand = os.path.join(root, name.replace("/", "-") + ".pt")
        if os.path.isfile(cand):
            return load(cand, device, jit, download_root, txt_length)
        warnings.warn(f"CLIP weights for {name} not found under {root} (no network): architecture built with "
                      f"random initialisation; call load_state_dict() to supply weights")
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
