"""Capture a golden vector of the REAL reference QualityNet (build container only) -> tests/golden/qnet.npz.

Imports ``/root/reference/models/qnet.py`` over the stub ``torchvision`` of ``oracle/_stub`` (own ResNet-18 with
torchvision's attribute names; the reference's branch wiring, pooling, merge and head are the reference's own
code).  Inputs are regenerated from Philox streams, weights from ``eva_vos_amd.synth.recipe_state_dict``.

Run:  python oracle/gen_golden_qnet.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "_stub"), "/root/reference", ROOT]
torch.set_grad_enabled(False)

from models.qnet import QualityNet as RefQNet  # noqa: E402  (reference)

from eva_vos_amd import synth  # noqa: E402
from eva_vos_amd.qnet import QualityNet  # noqa: E402


def inputs(n=3, seed=11):
    g = np.random.Generator(np.random.Philox(key=[seed, 224]))
    rgb = torch.from_numpy(g.normal(0, 1, (n, 3, 224, 224)).astype(np.float32))
    msk = torch.from_numpy((g.uniform(0, 1, (n, 1, 224, 224)) > 0.6).astype(np.float32)).expand(-1, 3, -1, -1)
    return rgb, msk.contiguous()


def main():
    mine = QualityNet().eval()
    sd = synth.recipe_state_dict(mine, seed=3)
    ref = RefQNet().eval()
    ref.load_state_dict(sd, strict=True)              # names and shapes equal the reference's
    mine.load_state_dict(sd, strict=True)
    rgb, msk = inputs()
    feats, logits = ref.extract_features(rgb, msk), ref(rgb, msk)
    print("features", tuple(feats.shape), "max |ref - own| =", float((feats - mine.extract_features(rgb, msk)).abs().max()),
          "scale", float(feats.abs().max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "qnet.npz"), features=feats.numpy(), logits=logits.numpy(),
                        names=np.array(list(ref.state_dict().keys())),
                        shapes=np.array([",".join(map(str, v.shape)) for v in ref.state_dict().values()]))


if __name__ == "__main__":
    main()
