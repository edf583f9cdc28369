"""GPU box: the backward sweep beside the forward one (STCN_DUAL_SWEEP, one video in flight) - A/B in ONE process on one box.
(a) BASELINE config 1's second interaction (T = 82: interact(0), then interact(41): both sweeps 40 frames, the backward one fused);
(b) an 8-round and a 60-round annotation session of the oracle mask policy on a resident clip (eva_vos_amd.eval_driver.run_policy).
python tools/dual_sweep_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import eval_driver, synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402
from mivos.inference_core import InferenceCore  # noqa: E402

torch.set_grad_enabled(False)
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop))
fuse.load_state_dict(synth.recipe_state_dict(fuse))
H, W = 480, 854


def r2(T, dual):
    os.environ["STCN_DUAL_SWEEP"] = dual
    img, gt = synth.synthetic_clip(T, H, W).cuda(), synth.synthetic_mask(T, H, W, 1)
    e = InferenceCore(prop, fuse, img, 1, engine_options={"lookahead": 2})
    e.interact(gt[:, 0], 0, download=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.interact(gt[:, T // 2], T // 2, download=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return e.stats()["frames"] / dt


def session(T, rounds, metric, dual):
    os.environ["STCN_DUAL_SWEEP"] = dual
    img, gt = synth.synthetic_clip(T, H, W, seed=7).cuda(), synth.synthetic_mask(T, H, W, 1, seed=7).cuda()
    smp = {"rgb": img, "gt": gt, "num_frames": T}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    core = InferenceCore(prop, fuse, img, 1, engine_options={"lookahead": 2})
    res = eval_driver.run_policy("oracle_mask", core, smp, rounds, metric)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return len(res["mu_metrics"]) / dt, res["propagated_frames"] / dt, res["frames"][:8]


r2(82, "1"); session(40, 8, "j", "1")                             # warm-up
for rep in range(3):
    for dual in ("0", "1"):
        print(f"rep {rep} STCN_DUAL_SWEEP={dual}: R2 of config 1 (T=82) {r2(82, dual):7.1f} frames/s | "
              + " | ".join(f"{T} frames x {r} rounds ({m}): %.1f rounds/s, %.0f frames/s %s" % session(T, r, m, dual) for T, r, m in ((40, 8, "j"), (66, 60, "j_and_f"))), flush=True)
