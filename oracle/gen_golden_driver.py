"""Pin the callers either side of the path to the reference (build container only) -> tests/golden/driver.npz:

* clip loader: ``datasets/annotation_dataset.py:80-132`` (``AnnotationDataset.__getitem__``) on the synthetic DAVIS-layout
  tree that ``eva_vos_amd.fq_driver.make_synthetic_tree`` writes (regenerated identically at test time);
* FQ-dataset writer: ``util/fq_dataset.py:26-91`` (``saver``: 224x224 nearest-resized mask PNGs + result rows;
  ``save_frames``: 224x224 bicubic-antialias RGB PNGs after a per-frame min-max normalisation).

torchvision is absent: the reference modules are imported over ``oracle/_stub/torchvision`` (ToTensor / Normalize /
Resize written from their public definitions; the tensor paths are the torch calls torchvision itself makes).

Run:  python oracle/gen_golden_driver.py
"""
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:] = [os.path.join(ROOT, "oracle", "_stub"), "/root/reference"] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
from datasets.annotation_dataset import AnnotationDataset  # noqa: E402  (reference)
from util import fq_dataset as RFQ  # noqa: E402  (reference)

sys.path.append(ROOT)
from eva_vos_amd import fq_driver  # noqa: E402

TREE = {"vidA": (4, 48, 64, 2), "vidB": (3, 60, 80, 1)}


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        imset = fq_driver.make_synthetic_tree(tmp, TREE)
        ds = AnnotationDataset(tmp, imset=imset)
        out["names"] = np.array(ds.videos)
        for i in range(len(ds)):
            d = ds[i]
            rgb, gt = d["rgb"], d["gt"]                       # [T,3,H,W] float32, [1,T,1,H,W] float32
            out[f"s{i}.rgb_shape"], out[f"s{i}.gt_shape"] = np.array(rgb.shape), np.array(gt.shape)
            out[f"s{i}.rgb_sample"] = rgb.reshape(-1)[::101].numpy().copy()
            out[f"s{i}.rgb_sum"] = np.array([rgb.double().sum().item(), rgb.double().abs().sum().item()])
            out[f"s{i}.gt"] = np.packbits(gt.numpy() > 0.5, axis=None)
            out[f"s{i}.num_frames"] = np.array(d["info"]["num_frames"])
        # writer: two states of vidA__1 (float masks [T,H,W] as interactions/eval.py hands them over) + the RGB frames
        d0 = ds[0]
        T, H, W = d0["rgb"].shape[0], d0["rgb"].shape[-2], d0["rgb"].shape[-1]
        rng = np.random.default_rng(3)
        yy, xx = np.mgrid[0:H, 0:W]
        gens = []
        for s in range(2):
            m = np.stack([(((yy - H * (0.4 + 0.05 * t)) / (0.3 * H)) ** 2 + ((xx - W * (0.45 + 0.04 * s)) / (0.25 * W)) ** 2 < 1)
                          ^ (rng.random((H, W)) < 0.01) for t in range(T)])
            gens.append(torch.from_numpy(m.astype(np.float32)))
        out["writer.gen"] = np.packbits(np.stack([g.numpy() > 0.5 for g in gens]), axis=None)
        out["writer.gen_shape"] = np.array([2, T, H, W])
        db = os.path.join(tmp, "FQ")
        ious = [[0.9, 0.2, 0.5, 0.7], [0.8, 0.6, 0.1, 20.0]]        # per-frame lists, as interactions/mask.py hands them over
        res = {"state_name": [], "ious": [], "selected_frame": []}
        nid, res = RFQ.saver(gens, [1, 2], ious, "vidA__1", 1, db, res, full_res=False, dont_save=[])
        out["writer.next_id"] = np.array(nid)
        out["writer.state_names"] = np.array(res["state_name"])
        out["writer.selected"] = np.array(res["selected_frame"])
        for sid in (1, 2):
            out[f"writer.masks{sid}"] = np.stack([np.array(Image.open(os.path.join(db, "Annotations", "224", f"vidA__1_round_{sid}", f"{t:05d}.png")))
                                                   for t in range(T)])
        RFQ.save_frames(d0["rgb"][None], "vidA", db, full_res=False)
        out["writer.rgb224"] = np.stack([np.array(Image.open(os.path.join(db, "RGBFrames", "224", "vidA", f"{t:05d}.png"))) for t in range(T)])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "driver.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
