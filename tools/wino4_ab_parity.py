"""GPU box: does the F(4x4) Winograd decoder cost mask parity?  BASELINE config 1 (480x854, k = 1, mem_freq = 5, T = 82,
interact(0) then interact(41)) on the CPU oracle once, and on the HIP engine in two fresh processes: default (decoder side on
F(4x4,3x3)) and STCN_WINO4=0 (F(2x2) / direct everywhere).  Prints the pixels differing from the oracle per arm and between the arms.
Usage: python tools/wino4_ab_parity.py [--frames 82]      (AB_ENV="STCN_WINO4_KEYPROJ=1": that arm instead of STCN_WINO4=0)"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 82
H, W, MF = 480, 854, 5


def inputs():
    from eva_vos_amd import synth
    return synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)


def nets():
    from eva_vos_amd import synth
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    p, f = PropagationNetwork(), FusionNet()
    psd, fsd = synth.recipe_state_dict(p), synth.recipe_state_dict(f)
    p.load_state_dict(psd)
    f.load_state_dict(fsd)
    return p, f, psd, fsd


def child(out):
    from mivos.inference_core import InferenceCore
    torch.set_grad_enabled(False)
    img, msk = inputs()
    p, f, _, _ = nets()
    core = InferenceCore(p, f, img.cuda(), 1, mem_freq=MF)
    r1 = core.interact(msk[:, 0], 0).copy()
    r2 = core.interact(msk[:, T // 2], T // 2).copy()
    np.savez_compressed(out, r1=r1, r2=r2)


def main():
    if "--child" in sys.argv:
        return child(sys.argv[sys.argv.index("--child") + 1])
    torch.set_grad_enabled(False)
    tmp = tempfile.mkdtemp()
    arms = {}
    arms_list = [("wino4", {}), ("no_wino4", {"STCN_WINO4": "0"})]
    if os.environ.get("AB_ENV"):                    # AB_ENV="NAME=VAL NAME=VAL": the second arm = the default engine under these variables
        arms_list = [("wino4", {}), ("no_wino4", dict(kv.split("=", 1) for kv in os.environ["AB_ENV"].split()))]
    elif os.environ.get("AB_LIB"):                    # a variant build as the second arm (STCN_LIB) instead of STCN_WINO4=0
        arms_list = [("wino4", {}), ("no_wino4", {"STCN_LIB": os.environ["AB_LIB"]})]
    for name, env in arms_list:
        out = os.path.join(tmp, name + ".npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--frames", str(T), "--child", out], env=dict(os.environ, **env))
        arms[name] = np.load(out)
    from oracle.stcn_oracle import OracleCore
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    img, msk = inputs()
    _, _, psd, fsd = nets()
    orc = OracleCore(psd, fsd, img, 1, mem_freq=MF)
    o1 = orc.interact(msk[:, 0], 0).copy()
    o2 = orc.interact(msk[:, T // 2], T // 2).copy()
    tot = o1.size
    for tag, ref in (("r1", o1), ("r2", o2)):
        row = []
        for name in arms:
            a = arms[name][tag] > 0
            b = ref > 0
            fu, fi = (a | b).reshape(T, -1).sum(1), (a & b).reshape(T, -1).sum(1)
            fiou = np.where(fu >= 64, fi / np.maximum(fu, 1), 1.0)
            row.append(f"{name}: {int((a != b).sum())} px differ from the CPU oracle (clip IoU {(a & b).sum() / max((a | b).sum(), 1):.6f}, worst frame {fiou.min():.6f} @ {int(fiou.argmin())})")
        ab = int(((arms['wino4'][tag] > 0) != (arms['no_wino4'][tag] > 0)).sum())
        extra = ""
        if "wino4_key" in arms:
            extra = f"; wino4_key differs from wino4 on {int(((arms['wino4_key'][tag] > 0) != (arms['wino4'][tag] > 0)).sum())} px"
        print(f"{tag} ({tot} px): " + "; ".join(row) + f"; the two arms differ from each other on {ab} px" + extra, flush=True)


if __name__ == "__main__":
    main()
