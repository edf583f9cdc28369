"""Host-side cost of returning the masks (GPU box): interact(..., download=True) vs download=False, T frames at 480x854."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import synth
from eva_vos_amd.params import FusionNet, PropagationNetwork
from mivos.inference_core import InferenceCore
torch.set_grad_enabled(False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 66
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop)); fuse.load_state_dict(synth.recipe_state_dict(fuse))
img = synth.synthetic_clip(T, 480, 854).cuda()
gt = synth.synthetic_mask(T, 480, 854, 1)
e = InferenceCore(prop, fuse, img, 1)
e.interact(gt[:, 0], 0)
for rep in range(3):
    for dl in (False, True):
        idx = (T // 2, T // 4, 3 * T // 4)[rep]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.interact(gt[:, idx], idx, download=dl)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"interact({idx}) download={dl}: {dt * 1e3:.1f} ms")
lw, uw, lh, uh = e.pad
out = e.masks[:, 0, lh:e.nh - uh, lw:e.nw - uw]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); a = out.cpu(); t1 = time.perf_counter(); b = a.numpy().astype(np.uint8); t2 = time.perf_counter()
    print(f".cpu() {1e3 * (t1 - t0):.2f} ms, .astype copy {1e3 * (t2 - t1):.2f} ms, {b.nbytes / 1e6:.1f} MB")
