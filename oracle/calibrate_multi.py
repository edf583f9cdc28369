"""Calibration of the MULTI-OBJECT weight recipe (eva_vos_amd/synth.py, RECIPES[2]) against the REAL reference.

Build container only (imports /root/reference through oracle/gen_golden.py's shims).  For a set of recipe knobs it runs the
reference InferenceCore on the config-3-shaped synthetic clip (k objects through the scribble path, every frame enters the bank)
and prints what the fixture `seq480k5` needs to be a statement about (almost) ALL pixels:

  * the fraction of pixels whose top-1 minus top-2 probability margin is >= 1e-2 in the reference's own output,
  * the size of every object's mask per frame (an object that vanishes makes its IoU vacuous),
  * with --noise: the reference against itself on 1 / 4 / 8 intra-op threads (per-object 1 - IoU, differing pixels).

  python oracle/calibrate_multi.py --size 240x432 --T 6 value_mask_gain=8 readout_gain=2 pred_gain=1 pred_bias=-3
The chosen constants are frozen in synth.RECIPES[2]; oracle/gen_golden.py --only=seq480k5 then writes the fixture.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (sets up the reference import path)

from eva_vos_amd import synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402


def load(knobs, seed=2):
    with contextlib.redirect_stdout(io.StringIO()):
        net, fus = G.RefNet().eval(), G.RefFus().eval()
    net.load_state_dict(synth.recipe_state_dict(PropagationNetwork(), seed, knobs), strict=True)
    fus.load_state_dict(synth.recipe_state_dict(FusionNet(), seed, knobs), strict=True)
    return net, fus


def run(net, fus, H, W, k, T, mem_freq, script, threads):
    torch.set_num_threads(threads)
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
    ref = G.RefCore(net, fus, img, k, mem_freq=mem_freq, device="cpu")
    outs = []
    for mf, idx in script:
        m = msk[:, mf]
        m = torch.cat([1 - m.sum(0, keepdim=True).clamp(0, 1), m], 0)
        rm = ref.interact(m.clone(), idx, scribble=True)
        lw, uw, lh, uh = ref.pad
        pr = ref.prob[:, :, 0, lh:ref.prob.shape[3] - uh if uh else None, lw:ref.prob.shape[4] - uw if uw else None].clone()
        outs.append((rm.copy(), pr))
    return outs


def main():
    args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    knobs = {a.split("=")[0]: float(a.split("=")[1]) for a in args}
    opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    H, W = (int(v) for v in opt.get("--size", "240x432").split("x"))
    T, k, mf = int(opt.get("--T", 6)), int(opt.get("--k", 5)), int(opt.get("--mem-freq", 1))
    script = [(0, 0)] + ([(T // 2, T // 2)] if "--two" in sys.argv else [])
    net, fus = load(knobs)
    print("knobs", {**synth.knobs_for(2), **knobs}, f"{H}x{W} T={T} k={k}", flush=True)
    t0 = time.time()
    base = run(net, fus, H, W, k, T, mf, script, 8)
    print(f"reference run: {time.time() - t0:.1f} s", flush=True)
    for r, (rm, pr) in enumerate(base):
        top = torch.topk(pr, 2, dim=0).values
        marg = (top[0] - top[1])
        print(f"round {r}: margin >= 1e-2 on {100 * float((marg >= 1e-2).float().mean()):.2f} % of pixels, >= 1e-3 on "
              f"{100 * float((marg >= 1e-3).float().mean()):.2f} %; per frame min {100 * float((marg >= 1e-2).float().mean((1, 2)).min()):.2f} %")
        areas = np.stack([(rm == o).reshape(T, -1).sum(1) for o in range(k + 1)], 0)
        print("  label areas per frame (rows = bg, objects):")
        for o in range(k + 1):
            print("   ", o, areas[o].tolist())
        nd = marg < 1e-2
        srt = torch.sort(pr, dim=0, descending=True).values
        if nd.any():
            t1, t3 = srt[0][nd], srt[2][nd]
            amx = pr.argmax(0)[nd]
            print(f"  non-decisive pixels: top-1 prob quartiles {np.quantile(t1.numpy(), [0.05, 0.25, 0.5, 0.75, 0.95]).round(3).tolist()}, "
                  f"3-way ties (top1 - top3 < 1e-2) {100 * float((t1 - t3 < 1e-2).float().mean()):.1f} %, bg is top-1 on {100 * float((amx == 0).float().mean()):.1f} %")
        pm = pr[1:].amax(0)
        print(f"  max object prob: mean {float(pm.mean()):.3f}; saturated (>1-1e-6) {100 * float((pr.amax(0) > 1 - 1e-6).float().mean()):.2f} %")
    if "--oracle" in sys.argv:          # the CPU oracle (another fp32 implementation: BN folded, sparse read-out) against the reference run
        from oracle import stcn_oracle as O
        torch.set_num_threads(8)
        seedk = {**synth.knobs_for(2), **knobs}
        orc = O.OracleCore(synth.recipe_state_dict(PropagationNetwork(), 2, seedk), synth.recipe_state_dict(FusionNet(), 2, seedk),
                           synth.synthetic_clip(T, H, W), k, mem_freq=mf)
        msk = synth.synthetic_mask(T, H, W, k)
        for r, (mfr, idx) in enumerate(script):
            m = msk[:, mfr]
            m = torch.cat([1 - m.sum(0, keepdim=True).clamp(0, 1), m], 0)
            om = orc.interact(m.clone(), idx, scribble=True)
            rm = base[r][0]
            per = []
            for o in range(1, k + 1):
                x, y = om == o, rm == o
                per.append(1 - (x & y).sum() / max((x | y).sum(), 1))
            print(f"round {r} oracle vs reference: {int((om != rm).sum())} mask pixels differ; per-object clip 1-IoU {[f'{v:.2e}' for v in per]}", flush=True)
    if "--noise" in sys.argv:
        runs = [base] + [run(net, fus, H, W, k, T, mf, script, nt) for nt in (1, 4)]
        for r in range(len(script)):
            worst, wf, npx = 0.0, 0.0, 0
            for i in range(3):
                for j in range(i + 1, 3):
                    a, b = runs[i][r][0], runs[j][r][0]
                    npx = max(npx, int((a != b).sum()))
                    for o in range(1, k + 1):
                        x, y = (a == o).reshape(T, -1), (b == o).reshape(T, -1)
                        u, n = (x | y).sum(), (x & y).sum()
                        if u:
                            worst = max(worst, 1 - n / u)
                        uf, nf = (x | y).sum(1), (x & y).sum(1)
                        ok = uf >= 64
                        if ok.any():
                            wf = max(wf, float((1 - nf[ok] / uf[ok]).max()))
            a, b, pr = runs[0][r][0], runs[1][r][0], runs[0][r][1]
            df = torch.from_numpy(a != b)
            if df.any():
                srt = torch.sort(pr, dim=0, descending=True).values
                m12, m13 = (srt[0] - srt[1])[df], (srt[0] - srt[2])[df]
                print(f"  pixels differing 8 vs 1 threads: {int(df.sum())}; margin top1-top2 quantiles {np.quantile(m12.numpy(), [0.5, 0.9, 1.0]).tolist()}, "
                      f"top1-top3 < 1e-4 on {100 * float((m13 < 1e-4).float().mean()):.0f} %, top-1 median {float(srt[0][df].median()):.3f}, "
                      f"bg involved {100 * float(((torch.from_numpy(a)[df] == 0) | (torch.from_numpy(b)[df] == 0)).float().mean()):.0f} %, per frame {df.reshape(T, -1).sum(1).tolist()}")
            print(f"round {r} self-noise (8/1/4 threads): worst per-object clip 1-IoU {worst:.2e}, worst frame {wf:.2e}, differing pixels {npx}")


if __name__ == "__main__":
    main()
