"""helper of tests/test_dropin.py: prints {symbol: [parameter names]} as JSON for the caller-facing symbols of the
Stage-1 path, imported through the reference's OWN import names.  argv[1] = 'reference' (the live /root/reference through
oracle/ref_shim.py) or 'dropin' (tris_amd.dropin.install())."""
import inspect
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TRIS_RANDOM_INIT", "1")
which = sys.argv[1]
if which == "reference":
    from oracle import ref_shim
    ref_shim.install_dataset()
else:
    import tris_amd.dropin
    tris_amd.dropin.install()

# the import statements of the reference's callers, verbatim (train_stage1.py:12-31, validate.py:12-24, model_stage1.py:1-12)
import CLIP.clip as clip                                   # noqa: E402
from args import get_parser                                # noqa: E402
from dataset.ReferDataset import ReferDataset              # noqa: E402
from dataset.transform import get_transform                # noqa: E402
from loss.clip_loss import clip_forward                    # noqa: E402
from model.attn import bilateral_prompt                    # noqa: E402
from model.model_stage1 import TRIS                        # noqa: E402
from utils.util import AverageMeter, compute_mask_IU, load_checkpoint, load_pretrained_checkpoint, save_checkpoint  # noqa: E402
if which == "dropin":
    from validate import isCorrectHit, validate, validate_same_sentence   # noqa: E402
    import train_stage1                                                     # noqa: E402
    extra = {"validate.validate": validate, "validate.validate_same_sentence": validate_same_sentence,
             "validate.isCorrectHit": isCorrectHit, "train_stage1.train_one_epoch": train_stage1.train_one_epoch,
             "train_stage1.clip_forward": train_stage1.clip_forward, "train_stage1.MaxLoss": train_stage1.MaxLoss}
else:   # validate.py / train_stage1.py of the reference import tensorboardX, cv2, logger ... at module level: read their
    import ast  # signatures from the source instead of importing them
    extra = {}
    for mod, names in (("validate", ("validate", "validate_same_sentence", "isCorrectHit")),
                       ("train_stage1", ("train_one_epoch", "clip_forward", "MaxLoss"))):
        tree = ast.parse(open(os.path.join("/root/reference", mod + ".py")).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                extra[f"{mod}.{node.name}"] = [a.arg for a in node.args.args]


def params(f):
    if isinstance(f, list):
        return f
    return [p for p in inspect.signature(f).parameters if p != "self"]


syms = {"TRIS.__init__": TRIS.__init__, "TRIS.forward": TRIS.forward, "TRIS.trainable_parameters": TRIS.trainable_parameters,
        "bilateral_prompt.__init__": bilateral_prompt.__init__, "bilateral_prompt.forward": bilateral_prompt.forward,
        "clip.load": clip.load, "clip.tokenize": clip.tokenize, "clip_forward": clip_forward,
        "get_transform": get_transform, "ReferDataset.__init__": ReferDataset.__init__,
        "AverageMeter.update": AverageMeter.update, "compute_mask_IU": compute_mask_IU,
        "save_checkpoint": save_checkpoint, "load_checkpoint": load_checkpoint,
        "load_pretrained_checkpoint": load_pretrained_checkpoint}
syms.update(extra)
out = {k: params(v) for k, v in syms.items()}
out["parser_dests"] = sorted(a.dest for a in get_parser()._actions)
out["TRIS.module"] = TRIS.__module__
out["clip.has_tokenizer"] = hasattr(clip, "_tokenizer") or hasattr(clip, "tokenize")
print("PROBE " + json.dumps(out))
