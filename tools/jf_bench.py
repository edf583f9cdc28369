"""Time the GPU J/F counts kernel on a 66-frame 480x854 clip (GPU box): python tools/jf_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import metrics, synth  # noqa: E402

T, H, W = 66, 480, 854
gt = (synth.synthetic_mask(T, H, W, 1)[0, :, 0] > 0.5).cuda()
pr = torch.roll(gt, shifts=(3, -5), dims=(1, 2))
metrics.sequence_scores_gpu(gt, pr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r = metrics.sequence_scores_gpu(gt, pr)
torch.cuda.synchronize()
print(f"J/F of {T} frames {H}x{W}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per call; mean J {r[:, 0].mean():.4f} F {r[:, 1].mean():.4f}")
