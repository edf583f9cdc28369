"""Fit `decoder.pred` of the MULTI-OBJECT weight recipe (synth.RECIPES[2]) on features of the REAL reference.

Why: with purely random weights the decoder answers every object of a frame with (nearly) the same logit, the soft aggregation
leaves most of a multi-object frame at p ~ 1/(k+1) in every row and the argmax there hangs on the last ulp - the REFERENCE does
not agree with itself across thread counts, and a parity statement could only be made on a minority of the pixels (round 3).
A random network with a FITTED LINEAR READ-OUT is the cheapest network that actually separates the objects: every weight of the
recipe stays a Philox draw except the 256 x 3 x 3 + 1 numbers of `decoder.pred`, which are the ridge least-squares solution that
maps the decoder's last feature map to +/- TARGET inside / outside each object's mask.

Procedure (build container only; imports /root/reference through oracle/gen_golden.py's shims):
  1. the reference network with the seed-2 Philox recipe walks the synthetic clips frame by frame with TEACHER FORCING - the
     memory holds values encoded from the true masks - and a forward hook collects the input of `decoder.pred` per object;
  2. normal equations over all (frame, object, pixel) rows in float64, ridge-regularised; the solution is rounded to fp32 and
     written to eva_vos_amd/recipe_data/pred_seed2.npz (11 KB, committed: data, not code);
  2b. --free-rounds more passes in FREE-RUNNING mode (memory values from the network's own output with the layer fitted so far,
     targets still the true masks) are added to the same normal equations: the layer learns to hold an object it has drifted on;
  3. oracle/calibrate_multi.py then runs the reference FREE (InferenceCore.interact) with the fitted layer and reports the
     decisive-pixel fraction, the object sizes and the reference's self-noise across thread counts.

  python oracle/fit_multi_pred.py --fusion       (afterwards: FusionNet.final_conv the same way, on sessions with a second interaction)
  python oracle/fit_multi_pred.py [--clips=480x854x12,480x854x104:8] [--target=5] [--ridge=1e-3] [--posw=10] [--free-rounds=2]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402
import calibrate_multi as CM  # noqa: E402

from eva_vos_amd import synth  # noqa: E402

OUT = os.path.join(G.ROOT, "eva_vos_amd", "recipe_data", "pred_seed2.npz")


def collect(net, H, W, T, k, frames, target, posw, free=False):
    """Walk over the first `frames` frames of the (T, H, W) synthetic clip; returns (A^T A, A^T y, weight sum) of the least-squares
    system over rows [unfolded 3x3 x 256 features | 1] at 1/4 scale.  free=False: teacher forcing (the memory values are encoded
    from the true masks); free=True: as InferenceCore.do_pass runs (inference_core.py:165-180) - the memory values of frame
    t > 0 come from the network's OWN aggregated output with the decoder.pred currently loaded, targets stay the true masks
    (the fit learns to recover from its own drift)."""
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
    imgs, _ = G.pad_divide_by(img, 16)
    feats = []
    hook = net.decoder.pred.register_forward_hook(lambda m, inp, out: feats.append(inp[0].detach()))
    keys, vals = [], []
    D = 256 * 9 + 1
    ata, aty, n = torch.zeros(D, D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64), 0
    for t in range(frames):
        k16, f16_thin, f16, f8, f4 = net.encode_key(imgs[:, t])
        m_t, _ = G.pad_divide_by(msk[:, t], 16)                                  # [k,1,nh,nw]
        if t > 0:
            feats.clear()
            prob = net.segment_with_query(torch.stack(keys, 2), torch.cat(vals, 2), f8, f4, k16, f16_thin)
            x = feats[0]                                                         # [k,256,h4,w4]  (ReLU already applied by the decoder)
            y = F.interpolate(m_t, size=x.shape[-2:], mode="area")[:, 0]         # [k,h4,w4] in [0,1]
            for o in range(k):
                a = F.unfold(x[o:o + 1], 3, padding=1)[0].t()                    # [h4*w4, 2304], (c, kh, kw) order = conv weight order
                a = torch.cat([a, torch.ones(a.shape[0], 1)], 1)
                yy = ((2 * y[o] - 1) * target).reshape(-1)
                # class balance: an object covers ~2 % of its frame; rows inside the mask weigh `posw` times a row outside
                wr = (1 + (posw - 1) * y[o].reshape(-1)).unsqueeze(1)
                # products of 8192-row chunks in fp32 (2305^2 x 1e5 rows x 55 (frame, object) pairs per pass in fp64 take an hour
                # on 8 cores), summed over the chunks in fp64
                for c0 in range(0, a.shape[0], 8192):
                    ac, wc = a[c0:c0 + 8192], wr[c0:c0 + 8192]
                    ata += (ac.t() @ (ac * wc)).double()
                    aty += (ac.t() @ (yy[c0:c0 + 8192] * wc[:, 0])).double()
                n += float(wr.sum())
        m_in = G.aggregate_wbg(prob, keep_bg=True)[1:] if (free and t > 0) else m_t
        vals.append(net.encode_value(imgs[:, t], f16, m_in))
        keys.append(k16)
    hook.remove()
    return ata, aty, n


FUS_OUT = os.path.join(G.ROOT, "eva_vos_amd", "recipe_data", "fusion_final_seed2.npz")


def collect_fusion(net, fus, H, W, T, k, mem_freq, script, target, posw, clip_seed=1, mask_seed=2):
    """A reference InferenceCore session (free-running) on a synthetic clip; every call of FusionNet (fuse_one_frame,
    inference_core.py:193-207: once per fused frame and object) contributes the rows [unfolded 3x3 x 32 features | 1] of its
    final_conv INPUT with the true mask of that frame and object as the target."""
    img, msk = synth.synthetic_clip(T, H, W, seed=clip_seed), synth.synthetic_mask(T, H, W, k, seed=mask_seed)
    ref = G.RefCore(net, fus, img, k, mem_freq=mem_freq, device="cpu")
    D = 32 * 9 + 1
    ata, aty, n = torch.zeros(D, D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64), 0.0
    state = {"ti": None, "obj": 0}
    orig = ref.fuse_one_frame

    def fuse_one_frame(tc, tr, ti, *a, **kw):
        state["ti"], state["obj"] = ti, 0
        return orig(tc, tr, ti, *a, **kw)

    ref.fuse_one_frame = fuse_one_frame

    def hook(mod, inp, out):
        nonlocal ata, aty, n
        o, ti = state["obj"], state["ti"]
        state["obj"] += 1
        x = inp[0].detach()                                                      # [1,32,nh,nw]
        y, _ = G.pad_divide_by(msk[o:o + 1, ti], 16)                             # [1,1,nh,nw]
        a = F.unfold(x, 3, padding=1)[0].t()
        a = torch.cat([a, torch.ones(a.shape[0], 1)], 1)
        yy = ((2 * y - 1) * target).reshape(-1)
        wr = (1 + (posw - 1) * y.reshape(-1)).unsqueeze(1)
        for c0 in range(0, a.shape[0], 65536):
            ac, wc = a[c0:c0 + 65536], wr[c0:c0 + 65536]
            ata += (ac.t() @ (ac * wc)).double()
            aty += (ac.t() @ (yy[c0:c0 + 65536] * wc[:, 0])).double()
        n += float(wr.sum())

    h = fus.final_conv.register_forward_hook(hook)
    for mf, idx in script:
        m = msk[:, mf]
        m = torch.cat([1 - m.sum(0, keepdim=True).clamp(0, 1), m], 0)
        ref.interact(m.clone(), idx, scribble=True)
    h.remove()
    return ata, aty, n


def fit_fusion(opt):
    """FusionNet.final_conv (32 x 3 x 3 + 1 numbers) of the multi-object recipe: with random weights the fused frames of a
    second interaction come out at p ~ 0.5 for every object (half the frame labelled, objects tied); fitted like decoder.pred."""
    target, ridge, posw = float(opt.get("--target", 5)), float(opt.get("--ridge", 1e-3)), float(opt.get("--posw", 10))
    torch.set_num_threads(8)
    net, fus = CM.load({"fusion_fitted": 0.0})
    D = 32 * 9 + 1
    ata, aty, n = torch.zeros(D, D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64), 0.0
    sessions = [dict(H=480, W=854, T=12, k=5, mem_freq=1, script=[(0, 0), (7, 7)]),
                dict(H=480, W=854, T=8, k=3, mem_freq=3, script=[(0, 0), (5, 5)], clip_seed=41, mask_seed=42),
                dict(H=240, W=432, T=10, k=2, mem_freq=2, script=[(0, 0), (6, 6), (3, 3)], clip_seed=5, mask_seed=6)]
    for it in range(2):                                   # pass 0: random final_conv (its input does not depend on it in round 2);
        for c in sessions:                                # pass 1: with the fitted layer (round-3 inputs see fused probabilities)
            t0 = time.time()
            a, b, m = collect_fusion(net, fus, target=target, posw=posw, **c)
            ata += a; aty += b; n += m
            print(f"fusion pass {it} session {c['H']}x{c['W']} T={c['T']} k={c['k']}: weight {m:.0f} in {time.time() - t0:.0f} s", flush=True)
        lam = ridge * float(torch.diagonal(ata)[:-1].mean())
        reg = torch.eye(D, dtype=torch.float64) * lam
        reg[-1, -1] = 0
        sol = torch.linalg.solve(ata + reg, aty)
        resid = float((sol @ ata @ sol - 2 * sol @ aty) / n + target ** 2)
        print(f"fusion pass {it}: weighted mean squared residual {resid:.3f}", flush=True)
        fus.final_conv.weight.copy_(sol[:-1].reshape(1, 32, 3, 3).float())
        fus.final_conv.bias.copy_(sol[-1:].float())
    np.savez(FUS_OUT, weight=sol[:-1].reshape(1, 32, 3, 3).float().numpy(), bias=np.array([float(sol[-1])], np.float32),
             meta=np.array([target, ridge, n, resid, posw]))
    print(f"fitted FusionNet.final_conv: |w| max {float(sol[:-1].abs().max()):.3f}, bias {float(sol[-1]):.3f} -> {FUS_OUT}")


def main():
    opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    if "--fusion" in sys.argv:
        return fit_fusion(opt)
    clips = opt.get("--clips", "480x854x12,480x854x104:8")
    target, ridge, k = float(opt.get("--target", 5)), float(opt.get("--ridge", 1e-3)), int(opt.get("--k", 5))
    posw = float(opt.get("--posw", 10))
    torch.set_num_threads(8)
    # the random pred of the recipe is irrelevant for the features (the hook reads pred's INPUT): fitted=False avoids loading a stale file
    net, _ = CM.load({"pred_fitted": 0.0})
    D = 256 * 9 + 1
    ata, aty, n = torch.zeros(D, D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64), 0
    def solve():
        lam = ridge * float(torch.diagonal(ata)[:-1].mean())
        reg = torch.eye(D, dtype=torch.float64) * lam
        reg[-1, -1] = 0
        sol = torch.linalg.solve(ata + reg, aty)
        return sol, float((sol @ ata @ sol - 2 * sol @ aty) / n + target ** 2)

    rounds = int(opt.get("--free-rounds", 2))       # passes in free-running mode after the teacher-forced one (normal equations accumulate)
    for it in range(1 + rounds):
        for c in clips.split(","):
            shape, _, fr = c.partition(":")
            H, W, T = (int(v) for v in shape.split("x"))
            t0 = time.time()
            a, b, m = collect(net, H, W, T, k, int(fr) if fr else T, target, posw, free=it > 0)
            ata += a; aty += b; n += m
            print(f"pass {it} ({'free' if it else 'teacher-forced'}) clip {c}: weight {m:.0f} in {time.time() - t0:.0f} s", flush=True)
        sol, resid = solve()
        print(f"pass {it}: weighted mean squared residual {resid:.3f}", flush=True)
        net.decoder.pred.weight.copy_(sol[:-1].reshape(1, 256, 3, 3).float())      # exactly the fp32 numbers that will be stored
        net.decoder.pred.bias.copy_(sol[-1:].float())
    w = sol[:-1].reshape(1, 256, 3, 3).float().numpy()
    b = np.array([float(sol[-1])], np.float32)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez(OUT, weight=w, bias=b, meta=np.array([target, ridge, n, resid, posw]))
    print(f"fitted decoder.pred on {n} rows: mean squared residual {resid:.3f} (target +/-{target}), |w| max {np.abs(w).max():.3f}, bias {b[0]:.3f} -> {OUT}")


if __name__ == "__main__":
    main()
