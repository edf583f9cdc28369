"""Rounds >= 2 of an annotation session on one clip (the regime the reference's loops spend their time in: 7 of the 8 rounds of
interactions/mask.py:113-146, 59 of 60 of eval_annotation_method.py:30): cached key features, memory read + decoder on every
frame, FusionNet + attention read on the frames between interacted frames.
Usage (GPU box): python tools/r2_profile.py [--frames T] [--rounds R]
  --rounds 1 runs the first interaction only: the kernel trace of `--rounds 8` minus that of `--rounds 1` is the trace of
  rounds 2..8 (tools/make_profile_summaries.py does the subtraction)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STCN_LOOKAHEAD", "0")        # clean per-class attribution by default; STCN_LOOKAHEAD=2 for the shipped mode
from eva_vos_amd import synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402
from mivos.inference_core import InferenceCore  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=66)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--no-class-profile", action="store_true", help="skip the second, HIP-event-profiled pass (for rocprofv3 runs)")
a = ap.parse_args()
torch.set_grad_enabled(False)
T = a.frames
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop))
fuse.load_state_dict(synth.recipe_state_dict(fuse))
img = synth.synthetic_clip(T, 480, 854).cuda()
gt = synth.synthetic_mask(T, 480, 854, 1)
order = [0, T // 2, T // 4, 3 * T // 4, T // 8, 3 * T // 8, 5 * T // 8, 7 * T // 8][:a.rounds]
for prof in ((False,) if a.no_class_profile else (False, True)):
    e = InferenceCore(prop, fuse, img, 1)
    acc, frames, secs = {}, 0, 0.0
    for r, idx in enumerate(order):
        e.set_profiling(prof and r > 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.interact(gt[:, idx], idx, download=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s = e.stats()
        print(f"round {r + 1} (frame {idx:2d}) profiling={prof}: {s['frames']} frames in {dt * 1e3:6.1f} ms = {s['frames'] / dt:7.1f} frames/s  {s}")
        if r > 0:
            frames += s["frames"]
            secs += dt
            if prof:
                for c, v in e.kernel_profile().items():
                    d = acc.setdefault(c, dict(ms=0.0, launches=0, flops=0.0, exec_flops=0.0))
                    for k in d:
                        d[k] += v[k]
    if frames:
        print(f"rounds 2..{len(order)} profiling={prof}: {frames} frames in {secs * 1e3:.1f} ms = {frames / secs:.1f} frames/s (one video in flight, no mask download)")
    if prof and acc:
        acc.pop("conv_hbm_bound", None)
        tot = sum(v["ms"] for v in acc.values())
        print(f"kernel time {tot:.1f} ms = {tot / frames:.3f} ms per propagated frame")
        for c, v in acc.items():
            if v["launches"]:
                print(f"  {c:12s} {v['ms']:8.2f} ms {100 * v['ms'] / tot:5.1f}%  launches {v['launches']:5d}  "
                      f"{v['flops'] / 1e9 / max(v['ms'], 1e-9):8.1f} TFLOP/s algorithmic  {v['exec_flops'] / 1e9 / max(v['ms'], 1e-9):8.1f} executed")
    del e
