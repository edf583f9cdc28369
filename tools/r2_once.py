"""interact(mask, 0) then interact(mask, T//2) on one clip, timed, three fresh engines (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import synth
from eva_vos_amd.params import FusionNet, PropagationNetwork
from mivos.inference_core import InferenceCore
torch.set_grad_enabled(False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 82
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop)); fuse.load_state_dict(synth.recipe_state_dict(fuse))
img = synth.synthetic_clip(T, 480, 854).cuda()
gt = synth.synthetic_mask(T, 480, 854, 1)
for rep in range(3):
    e = InferenceCore(prop, fuse, img, 1)
    out = []
    for idx in (0, T // 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = e.interact(gt[:, idx], idx)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.append(f"interact({idx}): {e.stats()['frames']} frames {dt * 1e3:.1f} ms = {e.stats()['frames'] / dt:.0f} frames/s")
    print(f"rep {rep}: " + "; ".join(out))
    del e
