"""GPU box: where does an annotation round of the drivers spend its time?  One resident 480p clip, the oracle mask policy
(interactions/mask.py:113-146) for R rounds, one lane.  Pass A: as the drivers run it (no extra syncs): wall per round.  Pass B: a
device sync after every segment (interact / metric / selection / state output): GPU + host time per segment.  Pass C: engine profiling
on (HIP events per launch): kernel ms per round = the device time the round NEEDS; wall - kernel ms = gaps.
python tools/session_breakdown.py [--frames 40] [--rounds 8] [--metric j|j_and_f]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import eval_driver, fq_driver, synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402
from mivos.inference_core import InferenceCore  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--metric", default="j")
ap.add_argument("--lookahead", type=int, default=2)
a = ap.parse_args()
torch.set_grad_enabled(False)
T, H, W = a.frames, 480, 854
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop))
fuse.load_state_dict(synth.recipe_state_dict(fuse))
img = synth.synthetic_clip(T, H, W, seed=7).cuda()
gt = synth.synthetic_mask(T, H, W, 1, seed=7)[0].cuda()             # [T,1,H,W]
gt_thw = gt[:, 0]
eo = {"lookahead": a.lookahead}


def session(mode):
    core = InferenceCore(prop, fuse, img, 1, engine_options=eo)
    core.set_profiling(mode == "prof")
    frames, rows = [0], []
    sync = torch.cuda.synchronize if mode in ("sync", "prof") else (lambda: None)
    for r in range(a.rounds):
        f = frames[r]
        t0 = time.perf_counter()
        core.interact(gt[f][None], f, download=False)
        sync()
        t1 = time.perf_counter()
        mu, gen, q = eval_driver.frame_quality(core, gt_thw, frames[:r + 1], a.metric)
        sync()
        t2 = time.perf_counter()
        sel = int(np.argmin(q))
        small = torch.nn.functional.interpolate(gen[:, None].float(), size=(224, 224), mode="nearest")[:, 0]
        host = (small * 255).to(torch.uint8).cpu().numpy()
        t3 = time.perf_counter()
        kms = sum(v["ms"] for c, v in core.kernel_profile().items() if c != "conv_hbm_bound") if mode == "prof" else 0.0
        rows.append((f, core.stats()["frames"], core.stats()["fused"], 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), kms, 1e3 * core.last_enqueue_s))
        frames.append(sel)
        del host
    return rows


session("plain")                                                       # warm-up
for mode in ("plain", "sync", "prof"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = session(mode)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fr = sum(r[1] for r in rows)
    print(f"--- mode {mode}: {a.rounds} rounds, {fr} propagated frames in {1e3 * dt:.1f} ms = {a.rounds / dt:.1f} rounds/s, {fr / dt:.0f} frames/s")
    print("  round frame frames fused | interact ms  metric ms  select+state ms | kernel ms  enqueue ms")
    for i, r in enumerate(rows):
        print(f"  {i + 1:5d} {r[0]:5d} {r[1]:6d} {r[2]:5d} | {r[3]:10.2f} {r[4]:10.2f} {r[5]:15.2f} | {r[6]:9.2f} {r[7]:10.2f}")
    if mode == "prof":
        k = sum(r[6] for r in rows)
        print(f"  kernel ms total {k:.1f} of wall {1e3 * dt:.1f} (profiled pass; engine kernels only)")
