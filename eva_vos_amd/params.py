"""Parameter containers for the STCN propagation network and the fusion CNN.

These classes hold *weights only*.  They reproduce the ``state_dict`` key
layout of the reference modules so the published checkpoints
(``stcn.pth`` / ``fusion.pth``) load with ``strict=True``:

* ``PropagationNetwork``  <- reference ``mivos/model/propagation/prop_net.py:140-151``
  (value encoder ``modules.py:93-124`` + ``mod_resnet.py:49-78,120-156``, key
  encoder = torchvision ResNet-50 stem..layer3 ``modules.py:127-149``, key
  projection ``modules.py:166-175``, ``key_comp`` ``prop_net.py:147``, decoder
  ``prop_net.py:13-30``; 405 tensors, 54 469 310 elements).
* ``FusionNet``           <- reference ``mivos/model/fusion_net.py:8-30`` (12 tensors).

There is deliberately no ``forward``: every computation on these weights is
done by the HIP engine (``csrc/``) which folds BatchNorm and repacks the
tensors once per model.  The trees are built from a compact spec instead of
hand-written module classes.
"""
from __future__ import annotations

import torch.nn as nn


class _Bag(nn.Module):
    """A pure container: named children, no forward."""

    def __init__(self, **children):
        super().__init__()
        for name, child in children.items():
            self.add_module(name, child)

    def forward(self, *a, **k):  # pragma: no cover - containers are not callable
        raise RuntimeError(
            "eva_vos_amd parameter containers have no forward(); use "
            "mivos.inference_core.InferenceCore (HIP engine) to run the network")


def _seq(*mods):
    """nn.Sequential keeps the integer child names ('0', '1', ...) the checkpoints use."""
    return nn.Sequential(*mods)


def _conv(cin, cout, k, stride=1, bias=True):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=bias)


# --------------------------------------------------------------------------- key encoder (ResNet-50)
def _bottleneck(cin, planes, stride, project):
    kids = dict(
        conv1=_conv(cin, planes, 1, bias=False), bn1=nn.BatchNorm2d(planes),
        conv2=_conv(planes, planes, 3, stride=stride, bias=False), bn2=nn.BatchNorm2d(planes),
        conv3=_conv(planes, planes * 4, 1, bias=False), bn3=nn.BatchNorm2d(planes * 4))
    if project:
        kids["downsample"] = _seq(_conv(cin, planes * 4, 1, stride=stride, bias=False),
                                  nn.BatchNorm2d(planes * 4))
    return _Bag(**kids)


def _r50_stage(cin, planes, nblocks, stride):
    blocks = [_bottleneck(cin, planes, stride, True)]
    blocks += [_bottleneck(planes * 4, planes, 1, False) for _ in range(nblocks - 1)]
    return _seq(*blocks)


def _key_encoder():
    return _Bag(conv1=_conv(3, 64, 7, stride=2, bias=False), bn1=nn.BatchNorm2d(64),
                res2=_r50_stage(64, 64, 3, 1),
                layer2=_r50_stage(256, 128, 4, 2),
                layer3=_r50_stage(512, 256, 6, 2))


# --------------------------------------------------------------------------- value encoder (ResNet-18 + fuser)
def _basic(cin, planes, stride):
    kids = dict(conv1=_conv(cin, planes, 3, stride=stride), bn1=nn.BatchNorm2d(planes),
                conv2=_conv(planes, planes, 3), bn2=nn.BatchNorm2d(planes))
    if stride != 1 or cin != planes:
        kids["downsample"] = _seq(_conv(cin, planes, 1, stride=stride), nn.BatchNorm2d(planes))
    return _Bag(**kids)


def _resblock(cin, cout):
    kids = dict(conv1=_conv(cin, cout, 3), conv2=_conv(cout, cout, 3))
    if cin != cout:
        kids["downsample"] = _conv(cin, cout, 3)
    return _Bag(**kids)


def _cbam(c, r=16):
    # mlp indices 1 and 3 are the Linear layers (0 = flatten, 2 = ReLU in the checkpoint numbering)
    mlp = _seq(nn.Identity(), nn.Linear(c, c // r), nn.Identity(), nn.Linear(c // r, c))
    return _Bag(ChannelGate=_Bag(mlp=mlp),
                SpatialGate=_Bag(spatial=_Bag(conv=_conv(2, 1, 7))))


def _value_encoder():
    return _Bag(conv1=_conv(5, 64, 7, stride=2), bn1=nn.BatchNorm2d(64),
                layer1=_seq(_basic(64, 64, 1), _basic(64, 64, 1)),
                layer2=_seq(_basic(64, 128, 2), _basic(128, 128, 1)),
                layer3=_seq(_basic(128, 256, 2), _basic(256, 256, 1)),
                fuser=_Bag(block1=_resblock(1024 + 256, 512), attention=_cbam(512),
                           block2=_resblock(512, 512)))


# --------------------------------------------------------------------------- decoder
def _upsample_block(skip_c, up_c, out_c):
    return _Bag(skip_conv=_conv(skip_c, up_c, 3), out_conv=_resblock(up_c, out_c))


def _decoder():
    return _Bag(compress=_resblock(1024, 512),
                up_16_8=_upsample_block(512, 512, 256),
                up_8_4=_upsample_block(256, 256, 256),
                pred=_conv(256, 1, 3))


class PropagationNetwork(_Bag):
    """Weights of the STCN propagation network (see module docstring)."""

    def __init__(self, top_k=50):
        super().__init__(value_encoder=_value_encoder(), key_encoder=_key_encoder(),
                         key_proj=_Bag(key_proj=_conv(1024, 64, 3)),
                         key_comp=_conv(1024, 512, 3),
                         decoder=_decoder())
        if top_k != 50:
            raise ValueError("the HIP memory reader is built for top_k=50 (reference prop_net.py:141)")
        self.top_k = top_k


class FusionNet(_Bag):
    """Weights of the 6-conv fusion CNN (reference fusion_net.py:12-30)."""

    def __init__(self):
        super().__init__(conv1=_seq(_conv(9, 32, 3)),
                         conv2=_seq(_conv(32, 32, 3), nn.Identity(), _conv(32, 32, 3)),
                         conv3=_seq(_conv(32, 32, 3), nn.Identity(), _conv(32, 32, 3)),
                         final_conv=_conv(32, 1, 3))
