"""Deterministic synthetic weights and clips (no checkpoints / datasets exist offline).

Everything is drawn from NumPy ``Philox`` streams keyed by ``(seed, crc32(name))`` so the build
container (where the goldens are captured from the reference) and the GPU box regenerate
bit-identical tensors.  The recipe is calibrated so that the network is
numerically *alive*: O(1) keys (the 50-way softmax has real spread), separated decoder logits
(few pixels near p=0.5), non-trivial BatchNorm statistics (so BN folding is really exercised).
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

# frozen calibration constants (see oracle/gen_golden.py --calibrate)
RES_GAMMA_SCALE = 0.3      # last BN gamma of every residual block
KEY_PROJ_GAIN = 0.09       # brings std(k16) to ~1
PRED_GAIN = 0.5            # decoder.pred weight gain -> logit std of a few units
PRED_BIAS = 0.0
FUSE_FINAL_GAIN = 6.0     # FusionNet final_conv gain (taps are made zero-sum like decoder.pred)


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(name.encode())]))


def _is_last_bn_of_block(name: str) -> bool:
    # ResNet-50 bottleneck: bn3 ; ResNet-18 basic block: bn2
    if name.startswith("key_encoder.") and ".bn3." in name:
        return True
    if name.startswith("value_encoder.layer") and ".bn2." in name:
        return True
    return False


def recipe_state_dict(module: torch.nn.Module, seed: int = 0) -> dict:
    """Return a full ``state_dict`` for ``module`` (PropagationNetwork or FusionNet container)."""
    out = {}
    for name, ref in module.state_dict().items():
        shape = tuple(ref.shape)
        g = _rng(seed, name)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(1, dtype=torch.int64)
            continue
        if name.endswith("running_mean"):
            a = g.normal(0.0, 0.1, shape)
        elif name.endswith("running_var"):
            a = g.uniform(0.5, 1.5, shape)
        elif ".bn" in name or ".downsample.1." in name:          # BatchNorm affine
            if name.endswith("weight"):
                a = g.uniform(0.5, 1.5, shape)
                if _is_last_bn_of_block(name):
                    a = a * RES_GAMMA_SCALE
            else:
                a = g.normal(0.0, 0.1, shape)
        elif name.endswith("bias"):
            a = g.normal(0.0, 0.02, shape)
        else:                                                    # conv / linear weight
            fan_in = int(np.prod(shape[1:]))
            a = g.normal(0.0, np.sqrt(2.0 / fan_in), shape)
            if name == "key_proj.key_proj.weight":
                a = a * KEY_PROJ_GAIN
            elif name == "decoder.pred.weight":
                # zero-sum taps per input channel: the logit then responds to spatial structure
                # only, so its mean stays ~0 at every resolution (mask neither all-fg nor all-bg)
                a = (a - a.mean(axis=(2, 3), keepdims=True)) * PRED_GAIN
            elif name == "final_conv.weight":
                a = (a - a.mean(axis=(2, 3), keepdims=True)) * FUSE_FINAL_GAIN
        if name == "decoder.pred.bias":
            a = a + PRED_BIAS
        out[name] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


def synthetic_clip(T: int, H: int, W: int, seed: int = 1) -> torch.Tensor:
    """Normalized-image-scale clip ``[1,T,3,H,W]``: smooth moving blobs + noise so consecutive
    frames are correlated (a pure-noise clip gives the memory reader nothing to match)."""
    g = _rng(seed, f"clip{T}x{H}x{W}")
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    base = g.normal(0.0, 1.0, (3, H, W)).astype(np.float32)
    frames = []
    for t in range(T):
        cy, cx = H * (0.45 + 0.10 * np.sin(0.21 * t)), W * (0.35 + 0.30 * t / max(T - 1, 1))
        blob = np.exp(-(((yy - cy) / (0.22 * H)) ** 2 + ((xx - cx) / (0.18 * W)) ** 2))
        f = 0.6 * base + np.stack([1.8 * blob - 0.5, 0.9 * blob, -1.2 * blob + 0.3])
        f = f + g.normal(0.0, 0.15, (3, H, W)).astype(np.float32)
        frames.append(f.astype(np.float32))
    return torch.from_numpy(np.stack(frames)[None])


def synthetic_mask(T: int, H: int, W: int, k: int = 1, seed: int = 2) -> torch.Tensor:
    """Ground-truth-like masks ``[k,T,1,H,W]`` (float 0/1, no bg channel): ellipses following the
    blobs of :func:`synthetic_clip`; objects are disjoint (stacked vertically)."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    m = np.zeros((k, T, 1, H, W), np.float32)
    for t in range(T):
        cy, cx = H * (0.45 + 0.10 * np.sin(0.21 * t)), W * (0.35 + 0.30 * t / max(T - 1, 1))
        for o in range(k):
            oy = cy + (o - (k - 1) / 2.0) * (0.36 * H / k)
            e = ((yy - oy) / (0.16 * H / k + 0.02 * H)) ** 2 + ((xx - cx) / (0.15 * W)) ** 2
            m[o, t, 0] = (e < 1.0)
    return torch.from_numpy(m)
