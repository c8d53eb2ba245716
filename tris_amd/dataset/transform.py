"""CPU image/target transforms with the reference's interface (dataset/transform.py:13-63), on PIL only.

The reference goes through torchvision's functional API on PIL inputs; what that does is: `F.resize(img, (s, s))` =
`img.resize((s, s), BILINEAR)`, `F.resize(target, ..., NEAREST)` likewise, `F.to_tensor` = uint8 HWC -> float32 CHW / 255,
`F.normalize` = (x - mean) / std.  This module is the host-side, per-sample path (and the definition the HBM pipeline in
tris_amd.dataset.hbm reproduces bit-for-bit on the GPU).
"""
import numpy as np
import torch
from PIL import Image

IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


class Resize(object):
    """image always to (size, size); the target only in training (evaluation scores at the original size)"""

    def __init__(self, output_size=384, train=True):
        self.size = output_size
        self.train = train

    def __call__(self, image, target):
        image = image.resize((self.size, self.size), Image.BILINEAR)
        if self.train:
            target = target.resize((self.size, self.size), Image.NEAREST)
        return image, target


class ToTensor(object):
    def __call__(self, image, target):
        arr = np.asarray(image)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        image = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        target = torch.tensor(np.asarray(target), dtype=torch.int64)
        return image, target


class Normalize(object):
    def __init__(self, mean, std):
        self.mean = mean
        self.std = std

    def __call__(self, image, target):
        mean = torch.as_tensor(self.mean, dtype=image.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=image.dtype).view(-1, 1, 1)
        return image.sub(mean).div_(std), target


def normalize_table(mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """float32 [3,256]: ToTensor+Normalize of every (channel, byte value), computed with the same float ops"""
    v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255).view(1, 256).repeat(len(mean), 1)
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1)
    return v.sub(m).div_(s)


def get_transform(size, train=True):
    return Compose([Resize(size, train), ToTensor(), Normalize(mean=IMAGENET_MEAN, std=IMAGENET_STD)])
