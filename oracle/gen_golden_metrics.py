"""Pin the J / F measures to the reference (build container only) -> tests/golden/metrics.npz.

Imports ``/root/reference/interactions/metrics.py``.  Its module-level imports need cv2, skimage.morphology and
torchmetrics, none of which is installed here, so three STAND-IN modules are registered first:

* ``_seg2bmap`` (metrics.py:38-97) is pure NumPy and runs as is: the boundary-map fixtures are the reference's own output.
* ``f_measure`` (metrics.py:100-160) calls ``cv2.dilate(bmap, disk(r))``.  Stand-ins, written from the libraries' public
  definitions: ``disk(r)`` = {(x, y): x^2 + y^2 <= r^2} on a (2r+1)^2 grid (skimage.morphology.disk), ``dilate`` = binary
  dilation by that structuring element anchored at its centre with nothing outside the image (cv2.dilate's default border).
  Everything else in f_measure (radius rule, special cases, precision / recall / F) is the reference's code.  The fixture
  keys say ``standin`` where a stand-in took part.
* ``JaccardIndex(task="binary")`` (torchmetrics): intersection / union, 0 for two empty masks - stand-in; the fixture's J
  values are marked the same way.

Run:  python oracle/gen_golden_metrics.py
"""
import os
import sys
import types

import numpy as np
import torch
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _install_standins():
    cv2 = types.ModuleType("cv2")
    cv2.dilate = lambda img, kernel: ndimage.binary_dilation(img.astype(bool), structure=kernel.astype(bool)).astype(np.uint8)
    sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.morphology")

    def disk(radius):
        r = int(radius)
        y, x = np.mgrid[-r:r + 1, -r:r + 1]
        return (x * x + y * y <= r * r).astype(np.uint8)

    skm.disk = disk
    sk.morphology = skm
    tm = types.ModuleType("torchmetrics")

    class JaccardIndex:
        def __init__(self, task="binary", num_classes=2):
            assert task == "binary"

        def __call__(self, a, b):
            a, b = a.bool(), b.bool()
            u = (a | b).sum()
            return (a & b).sum().float() / u if u > 0 else torch.tensor(0.0)

    tm.JaccardIndex = JaccardIndex
    sys.modules.update({"cv2": cv2, "skimage": sk, "skimage.morphology": skm, "torchmetrics": tm})


def masks(T, H, W, seed):
    """Moving ellipses with speckle, an empty frame, a full frame and objects touching every image edge."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    gt, pr = np.zeros((T, H, W), bool), np.zeros((T, H, W), bool)
    for t in range(T):
        cy, cx = H * (0.3 + 0.08 * t), W * (0.35 + 0.07 * t)
        gt[t] = ((yy - cy) / (0.2 * H)) ** 2 + ((xx - cx) / (0.25 * W)) ** 2 < 1
        pr[t] = ((yy - cy - 2) / (0.22 * H)) ** 2 + ((xx - cx + 3) / (0.2 * W)) ** 2 < 1
        pr[t] ^= rng.random((H, W)) < 0.003
    pr[0] = False                                   # n_fg == 0, n_gt > 0
    gt[1] = False                                   # n_gt == 0, n_fg > 0
    gt[2, :4, :] = True                             # touches the first rows
    gt[2, -6:, -9:] = True                          # touches the last row / column
    pr[2, :, :3] = True                             # touches the first columns
    if T > 4:
        gt[4] = pr[4] = False                       # both empty
    if T > 5:
        gt[5] = True                                # full frame
    return gt, pr


def main():
    _install_standins()
    sys.path.insert(0, "/root/reference")
    from interactions import metrics as R           # the reference module
    assert "/root/reference" in R.__file__
    out = {}
    for tag, (T, H, W) in {"small": (6, 60, 90), "odd": (4, 37, 53), "p480": (3, 480, 854)}.items():
        gt, pr = masks(T, H, W, seed=H)
        out[f"{tag}.gt"] = np.packbits(gt, axis=None)
        out[f"{tag}.pred"] = np.packbits(pr, axis=None)
        out[f"{tag}.shape"] = np.array([T, H, W])
        out[f"{tag}.bmap_gt"] = np.packbits(np.stack([R._seg2bmap(m.copy()) for m in gt]).astype(bool), axis=None)
        out[f"{tag}.bmap_pred"] = np.packbits(np.stack([R._seg2bmap(m.copy()) for m in pr]).astype(bool), axis=None)
        out[f"{tag}.f_standin"] = np.array([R.f_measure(gt[t], pr[t]) for t in range(T)], np.float64)
        out[f"{tag}.jf_standin"] = np.array([R.get_j_and_f(torch.from_numpy(gt[t:t + 1]), torch.from_numpy(pr[t:t + 1]))
                                             for t in range(T)], np.float64)
        out[f"{tag}.bound_pix"] = np.array(np.ceil(0.008 * np.linalg.norm((H, W))))
        print(tag, "F", out[f"{tag}.f_standin"].round(4).tolist(), "J&F", out[f"{tag}.jf_standin"].round(4).tolist())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics.npz"), **out)


if __name__ == "__main__":
    main()
