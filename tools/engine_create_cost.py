"""Cost of creating / destroying an engine (GPU box): InferenceCore(...) and del, 480x854, T frames."""
import os, sys, time
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
e = InferenceCore(prop, fuse, img, 1); e.interact(gt[:, 0], 0); del e      # model snapshot + warm-up
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = InferenceCore(prop, fuse, img, 1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    e.interact(gt[:, 0], 0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    del e
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"create {1e3 * (t1 - t0):.1f} ms, interact(0) {1e3 * (t2 - t1):.1f} ms, destroy {1e3 * (t3 - t2):.1f} ms")
print("torch allocated", torch.cuda.memory_allocated() / 1e9, "GB")
