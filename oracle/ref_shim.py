"""TEST INFRASTRUCTURE ONLY -- imports the *real* reference (fawnliu/TRIS at /root/reference)
inside the build container so that golden vectors can be generated from it.

Nothing here travels to the GPU box in any useful form: /root/reference does not exist
there, and `-m gpu` tests / bench.py / smoke() never import this module.  Only
`oracle/gen_golden.py` and the `not gpu` test that cross-checks the restatement against
the live reference (skipped when /root/reference is absent) use it.

The reference cannot be imported as-is (IDE junk imports of turtle/tkinter, missing
torchvision/ftfy/cv2/...).  We register empty stub modules for
those names and replace `clip.load` (which downloads weights) by a local constructor.
"""
import sys
import types

REF_ROOT = "/root/reference"

# CLIP constructor arguments (CLIP/clip/model.py:452-466 signature):
# (embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
#  context_length, txt_length, vocab_size, transformer_width, heads, layers)
CLIP_CTOR = {
    "RN50": (1024, 224, (3, 4, 6, 3), 64, None, 77, None, 49408, 512, 8, 12),
    "ViT-B/32": (512, 224, 12, 768, 32, 77, None, 49408, 512, 8, 12),
    "ViT-B-32": (512, 224, 12, 768, 32, 77, None, 49408, 512, 8, 12),
    "ViT-B/16": (512, 224, 12, 768, 16, 77, None, 49408, 512, 8, 12),
}


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


def install():
    """Make `import model.model_stage1`, `import CLIP.clip` ... resolve to the reference."""
    import os
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected on the build container only)")
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    _stub("turtle", forward=None)
    tk = _stub("tkinter", image_names=None)
    tk.__path__ = []
    _stub("tkinter.messagebox", NO=None)
    _stub("tkinter.tix", Tree=None)
    _stub("ftfy", fix_text=lambda s: s)
    tv = _stub("torchvision")
    tv.__path__ = []

    class _IM:   # values = Pillow's resampling constants, so the functional stub can hand them to Image.resize
        BICUBIC = 3
        BILINEAR = 2
        NEAREST = 0

    tvt = _stub("torchvision.transforms", Compose=_Dummy, Resize=_Dummy, CenterCrop=_Dummy,
                ToTensor=_Dummy, Normalize=_Dummy, InterpolationMode=_IM)
    tvt.__path__ = []
    _stub("torchvision.transforms.functional")
    _stub("termcolor", colored=lambda s, *a, **k: s)
    _stub("tensorboardX", SummaryWriter=_Dummy)
    _stub("cv2")
    _stub("imageio")

    import CLIP.clip as clip  # noqa: E402  (the reference's package)
    from CLIP.clip.model import CLIP  # noqa: E402

    def _load(name, device="cpu", jit=False, download_root=None, txt_length=77):
        a = list(CLIP_CTOR[name])
        a[6] = txt_length
        return CLIP(*a).float().eval(), None

    clip.load = _load
    clip.clip.load = _load
    return clip


def install_dataset():
    """Additionally make `dataset.ReferDataset` / `dataset.transform` of the reference importable.

    torchvision is absent, so the three functional ops the reference calls on PIL inputs are supplied here with
    torchvision 0.9's documented behaviour (transforms/functional_pil.py resize = Image.resize((w, h), interpolation);
    functional.to_tensor = uint8 HWC -> float CHW / 255; functional.normalize = (x - mean) / std).  pycocotools is
    absent too: `pycocotools.mask` is served by tris_amd.dataset.cocomask, so target masks are NOT independently pinned
    by anything that goes through this shim (cocomask's header says so)."""
    install()
    import numpy as np
    import torch

    tvf = sys.modules["torchvision.transforms.functional"]

    def resize(img, size, interpolation=2):
        return img.resize(tuple(size[::-1]), interpolation)

    def to_tensor(pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    def normalize(tensor, mean, std, inplace=False):
        t = tensor if inplace else tensor.clone()
        m = torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1)
        sd = torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
        return t.sub_(m).div_(sd)

    tvf.resize, tvf.to_tensor, tvf.normalize = resize, to_tensor, normalize
    sys.modules["torchvision.transforms"].functional = tvf
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    from tris_amd.dataset import cocomask
    pc = _stub("pycocotools")
    pc.__path__ = []
    pc.mask = _stub("pycocotools.mask", frPyObjects=cocomask.frPyObjects, decode=cocomask.decode, area=cocomask.area)
    sk = _stub("skimage")
    sk.__path__ = []
    sk.io = _stub("skimage.io")
    _stub("transformers")   # imported but unused by dataset/ReferDataset.py:12; the real one probes torchvision's spec
    from dataset.ReferDataset import ReferDataset   # noqa: E402  (the reference's)
    from dataset.transform import get_transform     # noqa: E402
    return ReferDataset, get_transform


def make_args(extra=()):
    install()
    from args import get_parser  # reference args.py
    base = ["--backbone", "clip-RN50", "--size", "320", "--max_query_len", "20",
            "--negative_samples", "3", "--batch_size", "2"]
    return get_parser().parse_args(base + list(extra))


def make_tris(extra=()):
    """Reference Stage-1 model (model/model_stage1.py:14) with stock random init."""
    install()
    from model.model_stage1 import TRIS
    return TRIS(make_args(extra))


def make_aux_clip(txt_length=20):
    """Reference aux CLIP ViT-B/32 (train_stage1.py:167), fp32 on CPU."""
    clip = install()
    m, _ = clip.load("ViT-B/32", txt_length=txt_length)
    return m
