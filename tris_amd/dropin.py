"""Drop-in under the reference's OWN import paths.

The reference's callers import the Stage-1 path as top-level modules (/root/reference/train_stage1.py:12-31,
validate.py:12-24, model/model_stage1.py:1-12, dataset/ReferDataset.py:1-20):

    from model.model_stage1 import TRIS          import CLIP.clip as clip            from validate import validate
    from model.attn import bilateral_prompt      from loss.clip_loss import clip_forward
    from args import get_parser                  from utils.util import AverageMeter, load_checkpoint, ...
    from dataset.ReferDataset import ReferDataset            from dataset.transform import get_transform

`install()` registers the tris_amd modules under exactly those names in `sys.modules`, so the reference's scripts run
on the MI355X path without editing a single import:

    import tris_amd.dropin; tris_amd.dropin.install()      # first line of train_stage1.py / validate.py
    from model.model_stage1 import TRIS                    # -> tris_amd.model.model_stage1.TRIS

It never shadows silently: if one of the names is already bound to a DIFFERENT module (e.g. the reference's own
`model` package was imported first) it raises unless `force=True`.  `uninstall()` removes exactly what install() added.
"""
import importlib
import sys

# reference import name -> tris_amd module
ALIASES = {
    "model": "tris_amd.model",
    "model.model_stage1": "tris_amd.model.model_stage1",
    "model.attn": "tris_amd.model.attn",
    "model.utils": "tris_amd.model.utils",
    "CLIP": "tris_amd.CLIP",
    "CLIP.clip": "tris_amd.CLIP.clip",
    "CLIP.clip.clip": "tris_amd.CLIP.clip.clip",
    "CLIP.clip.model": "tris_amd.CLIP.clip.model",
    "CLIP.clip.simple_tokenizer": "tris_amd.CLIP.clip.simple_tokenizer",
    "loss": "tris_amd.loss",
    "loss.clip_loss": "tris_amd.loss.clip_loss",
    "validate": "tris_amd.validate",
    "args": "tris_amd.args",
    "utils": "tris_amd.utils",
    "utils.util": "tris_amd.utils.util",
    "dataset": "tris_amd.dataset",
    "dataset.ReferDataset": "tris_amd.dataset.ReferDataset",
    "dataset.transform": "tris_amd.dataset.transform",
    "dataset.refer": "tris_amd.dataset.refer",
    "train_stage1": "tris_amd.train_stage1",
}

_installed = {}


def install(force=False):
    """Bind the reference's import names to the tris_amd modules.  Returns the {name: module} mapping."""
    mods = {name: importlib.import_module(target) for name, target in ALIASES.items()}
    clash = [n for n, m in mods.items() if n in sys.modules and sys.modules[n] is not m]
    if clash and not force:
        raise ImportError(f"tris_amd.dropin.install(): {clash} already imported from elsewhere "
                          f"({getattr(sys.modules[clash[0]], '__file__', '?')}); call install() before the reference's "
                          f"own modules are imported, or pass force=True")
    for name, m in mods.items():
        if sys.modules.get(name) is not m:
            _installed[name] = sys.modules.get(name)
            sys.modules[name] = m
    return mods


def uninstall():
    for name, prev in list(_installed.items()):
        if prev is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = prev
        del _installed[name]
