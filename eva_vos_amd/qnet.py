"""QualityNet + QNet frame selection on stock PyTorch-ROCm.

north_star keeps the QNet forward passes on PyTorch; what this module changes is everything around them:

* ``QualityNet``: own container of the reference's default configuration (``models/qnet.py:6-79``: two ResNet-18
  branches ``models/modules.py:12-62`` - rgb and 3x-repeated mask - 7x7 average pool, concatenation -> 1024 features,
  dropout + 20-way ``out_layer``); ``state_dict`` names equal the reference's, so ``qnet.pth`` loads strictly.
* ``qnet_frame_selection``: the reference (``interactions/policies.py:39-60`` with ``get_min_l2_dist`` ``:21-35``)
  moves every feature row to the host and runs an O(T * |interacted|) NumPy loop of ``np.linalg.norm`` calls per
  round; here the features stay on the device, one ``cdist`` gives all distances, and a single index crosses PCIe.
  Ties resolve like the reference's strict ``>`` scan: the first frame with the largest min-distance wins.
* 224x224 resizes (``policies.py:12-18``: torchvision ``Resize`` NEAREST for masks, BICUBIC + antialias for frames)
  are ``torch.nn.functional.interpolate`` calls on the device.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Block(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        else:
            self.downsample = None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class _Branch(nn.Module):
    """ResNet-18 trunk up to layer4 + 7x7 average pool -> [B,512,1,1] for 224x224 inputs."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        widths = (64, 128, 256, 512)
        cin = 64
        for i, w in enumerate(widths):
            setattr(self, f"layer{i + 1}", nn.Sequential(_Block(cin, w, 1 if i == 0 else 2), _Block(w, w, 1)))
            cin = w

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return F.avg_pool2d(x, 7)


class QualityNet(nn.Module):
    def __init__(self, n_labels: int = 20):
        super().__init__()
        self.rgb_branch = _Branch()
        self.mask_branch = _Branch()
        self.dropout = nn.Dropout(0.5)
        self.out_layer = nn.Linear(1024, n_labels)

    def extract_features(self, x_rgb, x_mask):
        return torch.cat((self.rgb_branch(x_rgb), self.mask_branch(x_mask)), 1).flatten(1)

    def forward(self, x_rgb, x_mask):
        return self.out_layer(self.dropout(self.extract_features(x_rgb, x_mask)))


def to_224(frames: torch.Tensor, masks: torch.Tensor):
    """frames [T,3,H,W] float, masks [T,H,W] (0/1) -> ([T,3,224,224], [T,3,224,224])."""
    imgs = F.interpolate(frames, size=(224, 224), mode="bicubic", antialias=True, align_corners=False)
    m = F.interpolate(masks[:, None].float(), size=(224, 224), mode="nearest")
    return imgs, m.expand(-1, 3, -1, -1)


def select_farthest(features: torch.Tensor, interacted) -> int:
    """argmax over frames of the min L2 distance to the interacted frames' features (first maximum wins)."""
    idx = torch.as_tensor(list(interacted), device=features.device, dtype=torch.long)
    d = torch.cdist(features[None].double(), features[idx][None].double())[0]        # [T, |interacted|]
    return int(torch.argmax(d.min(dim=1).values.float()).item())


@torch.no_grad()
def qnet_frame_selection(qnet, frames: torch.Tensor, masks: torch.Tensor, interacted, batch: int = 64) -> int:
    """frames [T,3,H,W], masks [T,H,W]; returns the selected frame index."""
    imgs, m3 = to_224(frames, masks)
    feats = torch.cat([qnet.extract_features(imgs[i:i + batch], m3[i:i + batch]) for i in range(0, len(imgs), batch)])
    return select_farthest(feats, interacted)
