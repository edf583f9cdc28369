"""End-to-end rate of the config-4 / config-5 drivers on a synthetic 480p dataset tree (GPU box):
python tools/driver_bench.py [videos] [frames]"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import eval_driver, fq_driver, synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402

torch.set_grad_enabled(False)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop))
fuse.load_state_dict(synth.recipe_state_dict(fuse))
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.perf_counter()
    imset = fq_driver.make_synthetic_tree(os.path.join(tmp, "db"), {f"v{i}": (T, 480, 854, 1) for i in range(nv)})
    print(f"dataset tree: {nv} videos x {T} frames 480x854 written in {time.perf_counter() - t0:.1f} s")
    fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "warm"), prop, fuse, rounds=1)      # warm-up (model upload)
    for name, fn in (("fq_driver (8 oracle rounds, PNG states + CSV)",
                      lambda: fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "fq"), prop, fuse, rounds=8)),
                     ("eval_driver oracle_mask (8 rounds, J&F)",
                      lambda: eval_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "e.csv"), prop, fuse, "oracle_mask", rounds=8))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rows = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: {nv} samples, {len(rows)} annotation rounds in {dt:.2f} s = {len(rows) / dt:.1f} rounds/s "
              f"({dt / nv:.2f} s per sample) incl. JPEG decode, H2D, propagation, metrics and output files")
