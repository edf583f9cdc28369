"""Per-kernel-class profile of a second interaction (cached key features, fusion on half the frames).
Usage (GPU box): python tools/r2_profile.py [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["STCN_LOOKAHEAD"] = "0"
from eva_vos_amd import synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402
from mivos.inference_core import InferenceCore  # noqa: E402

torch.set_grad_enabled(False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 66
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop))
fuse.load_state_dict(synth.recipe_state_dict(fuse))
img = synth.synthetic_clip(T, 480, 854).cuda()
gt = synth.synthetic_mask(T, 480, 854, 1)
for prof in (False, True):
    e = InferenceCore(prop, fuse, img, 1)
    e.interact(gt[:, 0], 0)
    e.set_profiling(prof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.interact(gt[:, T // 2], T // 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = e.stats()
    print(f"R2 profiling={prof}: {s['frames']} frames in {dt*1e3:.1f} ms = {s['frames']/dt:.1f} fps, stats {s}")
    if prof:
        kp = e.kernel_profile()
        tot = sum(v["ms"] for v in kp.values())
        for c, v in kp.items():
            if v["launches"]:
                print(f"  {c:12s} {v['ms']:8.2f} ms {100*v['ms']/tot:5.1f}%  launches {v['launches']:5d}  "
                      f"{v['flops']/1e9/max(v['ms'],1e-9):8.1f} TFLOP/s")
