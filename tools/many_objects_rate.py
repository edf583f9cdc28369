"""GPU box: propagation rate of ONE engine against the number of objects (480x854, T = 24, mem_freq = 5, scribble path): first interaction
and a fused second one, k = 1 .. STCN_MAX_OBJECTS.  python tools/many_objects_rate.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402
from mivos.inference_core import InferenceCore  # noqa: E402

torch.set_grad_enabled(False)
T, H, W = 24, 480, 854
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop, 2))
fuse.load_state_dict(synth.recipe_state_dict(fuse, 2))
img = synth.synthetic_clip(T, H, W).cuda()
print(f"{H}x{W}, T = {T}, mem_freq = 5, one engine (one video in flight); frames/s = propagated frames / wall time of interact(download=False) + sync")
for k in (1, 2, 5, 8, 9, 10, 16, 24, 32):
    msk = synth.synthetic_mask(T, H, W, k)
    rows = lambda f: torch.cat([1 - msk[:, f].sum(0, keepdim=True).clamp(0, 1), msk[:, f]], 0).cuda()      # noqa: E731
    m0, m1 = rows(0), rows(12)
    best = [1e9, 1e9]
    for rep in range(3):
        core = InferenceCore(prop, fuse, img, k, mem_freq=5)
        for i, (m, f) in enumerate(((m0, 0), (m1, 12))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            core.interact(m, f, scribble=True, download=False)
            torch.cuda.synchronize()
            best[i] = min(best[i], time.perf_counter() - t0)
        o = core.get_opts() if hasattr(core, "get_opts") else {}
        del core
    print(f"k = {k:2d}: first interaction {(T - 1) / best[0]:7.1f} frames/s ({1e3 * best[0] / (T - 1):.2f} ms per frame = {1e3 * best[0] / (T - 1) / k:.2f} per object), "
          f"fused second interaction {(T - 1) / best[1]:7.1f} frames/s; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GB (PyTorch side)")
