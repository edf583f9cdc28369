"""GPU box: WHICH pixels of BASELINE config 3 (480x854, k = 5, mem_freq = 1) does the HIP engine label differently from the REFERENCE
(tests/golden/long_cfg3.npz), and how decided are they?  First T frames (default 8): HIP engine, CPU oracle and the reference's label map.
Per frame and label pair (a -> b: the reference says a, the engine says b): the count, the engine's top-1 minus top-2 probability margin at
those pixels, the oracle's margin there, and whether the oracle sides with the reference.
python tools/cfg3_flip_probe.py [--frames 8]"""
import os
import sys
from collections import Counter

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402
from mivos.inference_core import InferenceCore  # noqa: E402
from oracle.stcn_oracle import OracleCore  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(min(32, os.cpu_count() or 1))
T = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 8
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "long_cfg3.npz"))
Tg, H, W, k, mf = (int(v) for v in g["shape"])
prop, fuse = PropagationNetwork(), FusionNet()
psd, fsd = synth.recipe_state_dict(prop, 2), synth.recipe_state_dict(fuse, 2)
prop.load_state_dict(psd)
fuse.load_state_dict(fsd)
img, msk = synth.synthetic_clip(Tg, H, W)[:, :T].contiguous(), synth.synthetic_mask(Tg, H, W, k)      # (the synthetic clip depends on its length: slice the full one)
m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
core = InferenceCore(prop, fuse, img.cuda(), k, mem_freq=mf)
a = core.interact(m0, 0, scribble=True)
orc = OracleCore(psd, fsd, img, k, mem_freq=mf)
o = orc.interact(m0.clone(), 0, scribble=True)
r = g["masks"][:T]
lw, uw, lh, uh = core.pad
crop = lambda p: p[:, :, 0, lh:p.shape[3] - uh if uh else None, lw:p.shape[4] - uw if uw else None]      # noqa: E731
ph, po = crop(core.prob.cpu()).numpy(), crop(orc.prob).numpy()          # [k+1, T, H, W]


def margin(p):
    s = np.sort(p, axis=0)
    return s[-1] - s[-2]


mh, mo = margin(ph), margin(po)
print(f"first {T} frames: HIP vs reference {int((a != r).sum())} px, oracle vs reference {int((o != r).sum())} px, HIP vs oracle {int((a != o).sum())} px")
for t in range(1, T):
    d = a[t] != r[t]
    pairs = Counter(zip(r[t][d].tolist(), a[t][d].tolist()))
    side = int((o[t][d] == r[t][d]).sum())
    print(f"frame {t}: {int(d.sum())} px differ from the reference (the oracle sides with the reference on {side} of them); label pairs reference->engine {dict(pairs.most_common(6))}; "
          f"engine margin at them: median {np.median(mh[t][d]) if d.any() else 0:.1e} max {mh[t][d].max() if d.any() else 0:.1e}; oracle margin: median {np.median(mo[t][d]) if d.any() else 0:.1e}; "
          f"|p_engine - p_oracle| at them: max over rows median {np.median(np.abs(ph[:, t][:, d] - po[:, t][:, d]).max(0)) if d.any() else 0:.1e}; "
          f"whole frame: |dp| p99.9 {np.quantile(np.abs(ph[:, t] - po[:, t]).max(0), 0.999):.1e}, pixels with engine margin < 1e-4: {int((mh[t] < 1e-4).sum())}, < 1e-3: {int((mh[t] < 1e-3).sum())}")
