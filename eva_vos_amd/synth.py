"""Deterministic synthetic weights and clips (no checkpoints / datasets exist offline).

Everything is drawn from NumPy ``Philox`` streams keyed by ``(seed, crc32(name))`` so the build
container (where the goldens are captured from the reference) and the GPU box regenerate
bit-identical tensors.  The recipe is calibrated so that the network is
numerically *alive*: O(1) keys (the 50-way softmax has real spread), separated decoder logits
(few pixels near p=0.5), non-trivial BatchNorm statistics (so BN folding is really exercised).
"""
from __future__ import annotations

import os
import zlib

import numpy as np
import torch

# frozen calibration constants (see oracle/gen_golden.py --calibrate)
RES_GAMMA_SCALE = 0.3      # last BN gamma of every residual block
KEY_PROJ_GAIN = 0.09       # brings std(k16) to ~1
PRED_GAIN = 0.5            # decoder.pred weight gain -> logit std of a few units
PRED_BIAS = 0.0
FUSE_FINAL_GAIN = 6.0     # FusionNet final_conv gain (taps are made zero-sum like decoder.pred)

# Per-seed recipe knobs.  Seeds 0 and 1 are the single-object recipe above (two independent draws).  Seed 2 is the MULTI-OBJECT
# recipe (oracle/calibrate_multi.py, frozen): with the plain recipe the value encoder hardly looks at its mask channels, the
# decoder answers every object of a frame with (nearly) the same logit and the soft aggregation leaves most of the frame at
# p ~ 1/(k+1) in every row - an argmax decided by the last ulp, on which the REFERENCE does not agree with itself across thread
# counts.  The multi-object recipe makes the memory values depend on the object mask (value_mask_gain on the two mask channels
# of value_encoder.conv1), lets the decoder listen to the read-out (readout_gain on the read-out half of decoder.compress) and
# gives decoder.pred a negative bias, so that where no object stands out the BACKGROUND wins by a wide margin: the reference's
# own top-1 minus top-2 probability margin is >= 1e-2 on > 90 % of the pixels of a 480p five-object clip (tests/golden/seq480k5).
DEFAULT_KNOBS = dict(res_gamma_scale=RES_GAMMA_SCALE, key_proj_gain=KEY_PROJ_GAIN, pred_gain=PRED_GAIN, pred_bias=PRED_BIAS,
                     fuse_final_gain=FUSE_FINAL_GAIN, value_mask_gain=1.0, readout_gain=1.0, frame_gain=1.0, value_f16_gain=1.0, value_others_gain=1.0, pred_fitted=0.0, fusion_fitted=0.0)
RECIPES = {2: dict(DEFAULT_KNOBS, value_mask_gain=8.0, value_others_gain=8.0, pred_fitted=1.0, fusion_fitted=1.0)}     # seed -> knobs (other seeds: DEFAULT_KNOBS)


def knobs_for(seed: int) -> dict:
    return dict(RECIPES.get(seed, DEFAULT_KNOBS))


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(name.encode())]))


def _is_last_bn_of_block(name: str) -> bool:
    # ResNet-50 bottleneck: bn3 ; ResNet-18 basic block: bn2
    if name.startswith("key_encoder.") and ".bn3." in name:
        return True
    if name.startswith("value_encoder.layer") and ".bn2." in name:
        return True
    return False


def recipe_state_dict(module: torch.nn.Module, seed: int = 0, knobs: dict | None = None) -> dict:
    """Return a full ``state_dict`` for ``module`` (PropagationNetwork or FusionNet container).  ``knobs`` overrides the
    frozen per-seed recipe constants (calibration only)."""
    kn = knobs_for(seed)
    kn.update(knobs or {})
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
                    a = a * kn["res_gamma_scale"]
            else:
                a = g.normal(0.0, 0.1, shape)
        elif name.endswith("bias"):
            a = g.normal(0.0, 0.02, shape)
        else:                                                    # conv / linear weight
            fan_in = int(np.prod(shape[1:]))
            a = g.normal(0.0, np.sqrt(2.0 / fan_in), shape)
            if name == "key_proj.key_proj.weight":
                a = a * kn["key_proj_gain"]
            elif name == "decoder.pred.weight":
                # zero-sum taps per input channel: the logit then responds to spatial structure
                # only, so its mean stays ~0 at every resolution (mask neither all-fg nor all-bg)
                a = (a - a.mean(axis=(2, 3), keepdims=True)) * kn["pred_gain"]
            elif name == "value_encoder.conv1.weight":
                a[:, 3] *= kn["value_mask_gain"]                 # channel 3: the object's own mask
                a[:, 4] *= kn["value_others_gain"]               # channel 4: the other objects' masks (identical for every non-owner)
            elif name in ("value_encoder.fuser.block1.conv1.weight", "value_encoder.fuser.block1.downsample.weight"):
                a[:, 256:] *= kn["value_f16_gain"]               # input channels [256, 1280): the key encoder's f16, the same for every object
            elif name in ("decoder.compress.conv1.weight", "decoder.compress.downsample.weight"):
                a[:, :512] *= kn["readout_gain"]                 # input channels [0, 512): the memory read-out (prop_net.py:189)
                a[:, 512:] *= kn["frame_gain"]                   # [512, 1024): f16_thin, the same for every object
            elif name in ("decoder.up_16_8.skip_conv.weight", "decoder.up_8_4.skip_conv.weight"):
                a = a * kn["frame_gain"]                         # skip connections: frame features, the same for every object
            elif name == "final_conv.weight":
                a = (a - a.mean(axis=(2, 3), keepdims=True)) * kn["fuse_final_gain"]
        if name == "decoder.pred.bias":
            a = a + kn["pred_bias"]
        if kn["pred_fitted"] and name in ("decoder.pred.weight", "decoder.pred.bias"):
            # the fitted linear read-out of the multi-object recipe (oracle/fit_multi_pred.py; 256 x 3 x 3 + 1 numbers, committed data)
            fit = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "recipe_data", f"pred_seed{seed}.npz"))
            a = fit["weight" if name.endswith("weight") else "bias"].astype(np.float64).reshape(shape)
        if kn["fusion_fitted"] and name in ("final_conv.weight", "final_conv.bias"):
            # FusionNet's fitted last layer of the multi-object recipe (oracle/fit_multi_pred.py --fusion; 32 x 3 x 3 + 1 numbers)
            fit = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "recipe_data", f"fusion_final_seed{seed}.npz"))
            a = fit["weight" if name.endswith("weight") else "bias"].astype(np.float64).reshape(shape)
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
