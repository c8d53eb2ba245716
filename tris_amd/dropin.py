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

One torch entry point is wrapped as well, because the reference calls it on the model and it would otherwise do NOTHING:
`nn.SyncBatchNorm.convert_sync_batchnorm(model)` (/root/reference/train_stage1.py:69) only converts `nn.BatchNorm*`
instances, and the BatchNorm of this path is `tris_amd.CLIP.clip.model.BatchNorm2d` (its own `nn.Module`: the statistics
exchange lives in ops.BatchNormFn).  Left alone, the reference's script would train data-parallel with UNSYNCHRONISED
statistics and no warning.  With the drop-in installed the call also sets `process_group` on every tris_amd BatchNorm2d
(= tris_amd.parallel.convert_sync_batchnorm) -- or raises if no process group exists to synchronise over.
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
_torch_patch = {}


def _patch_sync_batchnorm():
    import torch
    if "convert" in _torch_patch:
        return
    orig = torch.nn.SyncBatchNorm.__dict__["convert_sync_batchnorm"]   # the classmethod object itself

    def convert_sync_batchnorm(cls, module, process_group=None):
        out = orig.__func__(cls, module, process_group)
        from .CLIP.clip.model import BatchNorm2d
        if any(isinstance(m, BatchNorm2d) for m in out.modules()):
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("nn.SyncBatchNorm.convert_sync_batchnorm on a tris_amd model needs an initialised process "
                                   "group (torch.distributed.init_process_group): there is nothing to synchronise the "
                                   "BatchNorm statistics over")
            from .parallel import convert_sync_batchnorm as ours
            ours(out, process_group)
        return out
    _torch_patch["convert"] = orig
    torch.nn.SyncBatchNorm.convert_sync_batchnorm = classmethod(convert_sync_batchnorm)


def _unpatch_sync_batchnorm():
    import torch
    orig = _torch_patch.pop("convert", None)
    if orig is not None:
        torch.nn.SyncBatchNorm.convert_sync_batchnorm = orig


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
    _patch_sync_batchnorm()
    return mods


def uninstall():
    _unpatch_sync_batchnorm()
    for name, prev in list(_installed.items()):
        if prev is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = prev
        del _installed[name]
