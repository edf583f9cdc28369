"""GPU box: BASELINE config 3 at full length (480x854, k = 5, mem_freq = 1, T = 104) on the HIP engine against the label map the REFERENCE
produced (tests/golden/long_cfg3.npz), under the current environment - run once per arm (STCN_WINO4=0, STCN_WINO4_KEY=0 ...: model-creation
knobs are read once per process).  Prints per object the clip 1-IoU and the worst frame, and the frames where the differing pixels sit."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402
from mivos.inference_core import InferenceCore  # noqa: E402

torch.set_grad_enabled(False)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "long_cfg3.npz"))
T, H, W, k, mf = (int(v) for v in g["shape"])
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop, 2))
fuse.load_state_dict(synth.recipe_state_dict(fuse, 2))
img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
a = InferenceCore(prop, fuse, img.cuda(), k, mem_freq=mf).interact(m0, 0, scribble=True)
b = g["masks"]
arm = " ".join(f"{n}={os.environ[n]}" for n in sorted(os.environ) if n.startswith("STCN_")) or "defaults"
print(f"[{arm}] {int((a != b).sum())} of {a.size} px differ from the reference")
for o in range(1, k + 1):
    x, y = (a == o).reshape(T, -1), (b == o).reshape(T, -1)
    u, n = (x | y).sum(1), (x & y).sum(1)
    miss = np.where(u >= 64, 1 - n / np.maximum(u, 1), 0.0)
    d = (x != y).sum(1)
    first = int(np.argmax(d > 16)) if (d > 16).any() else -1
    print(f"  object {o}: clip 1-IoU {1 - n.sum() / u.sum():.2e}, worst frame {int(miss.argmax())}: {miss.max():.2e}; differing px {int(d.sum())}, first frame with > 16 of them: {first}; "
          f"px per quarter of the clip {[int(d[q * T // 4:(q + 1) * T // 4].sum()) for q in range(4)]}")
