"""GPU box: rate of the config-4 / config-5 drivers against the number of videos in flight per GPU (lanes).
python tools/driver_lanes.py [videos=8] [frames=40]"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import eval_driver, fq_driver, synth  # noqa: E402
from eva_vos_amd import inference_core as IC  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402

torch.set_grad_enabled(False)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop))
fuse.load_state_dict(synth.recipe_state_dict(fuse))
with tempfile.TemporaryDirectory() as tmp:
    imset = fq_driver.make_synthetic_tree(os.path.join(tmp, "db"), {f"v{i}": (T, 480, 854, 1) for i in range(nv)})
    fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "warm"), prop, fuse, rounds=2)
    for lanes in (1, 2, 3, 4):
        fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "warm"), prop, fuse, rounds=2, lanes=lanes)      # pool / allocator warm for THIS lane count
        for name, fn in (("fq_driver", lambda st: fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, f"fq{lanes}"), prop, fuse, rounds=8, lanes=lanes, stats=st)),
                         ("eval_driver", lambda st: eval_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "e.csv"), prop, fuse, "oracle_mask", rounds=8, lanes=lanes, stats=st))):
            torch.cuda.synchronize()
            st = {}
            IC.CREATE_ACCOUNT = {}
            t0 = time.perf_counter()
            rows = fn(st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"lanes {lanes} {name}: {len(rows)} rounds in {dt:.2f} s = {len(rows) / dt:.1f} rounds/s = {st.get('propagated_frames', 0) / dt:.0f} propagated frames/s; "
                  f"host account {st}; InferenceCore() phases {({k: round(v, 3) for k, v in IC.CREATE_ACCOUNT.items()})}", flush=True)
