import os, sys, tempfile, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eva_vos_amd import fq_driver, synth
from eva_vos_amd.params import FusionNet, PropagationNetwork
torch.set_grad_enabled(False)
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop)); fuse.load_state_dict(synth.recipe_state_dict(fuse))
nv, T = 8, 40
with tempfile.TemporaryDirectory() as tmp:
    imset = fq_driver.make_synthetic_tree(os.path.join(tmp, "db"), {f"v{i}": (T, 480, 854, 1) for i in range(nv)})
    fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "warm"), prop, fuse, rounds=2)
    for tag, kw in (("save_masks=True", dict(save_masks=True)), ("save_masks=False", dict(save_masks=False))):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rows = fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, f"o{tag}{rep}"), prop, fuse, rounds=8, lanes=2, **kw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{tag}: {len(rows)} rounds in {dt:.2f} s = {len(rows) / dt:.1f} rounds/s", flush=True)
