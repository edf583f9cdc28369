"""Stand-in for the few torchvision.transforms the reference's dataset / writer code uses (torchvision is not installed in
the build container).  Written from the public definitions; the tensor code paths are the very torch calls torchvision
makes for tensor inputs.  Test infrastructure: only oracle/gen_golden_*.py import the reference through it."""
import enum

import numpy as np
import torch
import torch.nn.functional as F


class InterpolationMode(enum.Enum):
    NEAREST = "nearest"
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"


class Compose:
    def __init__(self, ts):
        self.ts = list(ts)

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class ToTensor:
    """PIL image (uint8 HWC) -> float32 CHW in [0, 1]: img.permute(2, 0, 1).float().div(255)."""

    def __call__(self, pic):
        a = torch.from_numpy(np.array(pic, np.uint8, copy=True))
        if a.dim() == 2:
            a = a[:, :, None]
        return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)

    def __call__(self, x):
        return (x - self.mean[:, None, None]) / self.std[:, None, None]


class Resize:
    """Tensor path of torchvision.transforms.functional.resize: torch.nn.functional.interpolate on [..., H, W]
    (nearest: no align_corners / antialias; bicubic: align_corners=False, antialias as given, uint8-free float path)."""

    def __init__(self, size, interpolation=InterpolationMode.BILINEAR, antialias=None):
        self.size = tuple(size) if len(size) == 2 else (size[0], size[0])
        self.mode, self.antialias = interpolation, bool(antialias)

    def __call__(self, x):
        sq = x.dim() == 3
        x4 = x[None] if sq else x
        if self.mode == InterpolationMode.NEAREST:
            y = F.interpolate(x4.float(), size=self.size, mode="nearest")
        else:
            y = F.interpolate(x4.float(), size=self.size, mode=self.mode.value, align_corners=False, antialias=self.antialias)
        return y[0] if sq else y
