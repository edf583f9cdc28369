"""Annotation-policy evaluation on the HIP engine, sharded over the GPUs of one node (BASELINE config 5, mask policies).

Own counterpart of the reference driver ``eval_annotation_method.py:118-190`` for the policies whose annotations are
ground-truth masks - ``oracle_mask``, ``rand_mask``, ``qnet_mask``, ``upper_bound_mask`` (``interactions/mask.py:10-39,
42-71,74-103,196-227``) - with their helpers ``initialize`` / ``not_avail_frames`` / ``eval_processor_metric``
(``interactions/eval.py:27-117``) and the frame selectors of ``interactions/policies.py``.  The click / bbox policies
and ``eva_vos`` itself additionally need SAM (``segment_anything``) and the PPO agent, which north_star leaves on stock
PyTorch-ROCm and which are not installed offline; they call the same ``InferenceCore.interact`` boundary.

Differences by design:

* J / J&F per frame come from the device (``stcn_metrics_jf_counts``) - masks never visit the host per round;
* QNet frame selection keeps features on the device (``eva_vos_amd.qnet``);
* two samples are in flight per GPU (host thread + HIP stream each, ``--lanes``): +9..17 % rounds/s;
* samples are LPT-sharded over ranks instead of ``--min-idx/--max-idx``; ONE gather of fixed-width rows at the end
  (RCCL over xGMI), rank 0 writes the CSV with the reference's columns ``video, mu_metric, annotation_time, round``.

Usage:  python -m eva_vos_amd.eval_driver --root data/MOSE --imset data/MOSE/ImageSets/test.txt --policy oracle_mask
        (multi-GPU: python -m torch.distributed.run --nproc-per-node N -m eva_vos_amd.eval_driver ...)
"""
from __future__ import annotations

import argparse
import copy
import csv
import os
import random
import time
from typing import List

import numpy as np
import torch

from . import metrics, shard
from .fq_driver import NO_OBJECT, ClipDataset, lane_engine_options, run_lanes

POLICIES = ("oracle_mask", "rand_mask", "qnet_mask", "upper_bound_mask")
MASK_SECONDS, SKIP_SECONDS = 80, 3            # annotation cost model of interactions/mask.py:33-36


def frame_quality(processor, gt_thw: torch.Tensor, interacted: List[int], metric: str = "j_and_f"):
    """``eval_processor_metric`` (interactions/eval.py:27-81) for mask annotations: per-frame J or J&F of the engine's
    masks against the ground truth, annotated frames counting with their GT mask, NO_OBJECT token for empty GT.
    Returns (mean over frames with an object, generated masks uint8 [T,H,W] on the device, quality[T])."""
    lw, uw, lh, uh = processor.pad
    seg = processor.masks[:, 0, lh:processor.nh - uh, lw:processor.nw - uw] > 0
    gtb = gt_thw > 0.5
    gen = seg.clone()
    if interacted:
        gen[interacted] = gtb[interacted]
    rows = metrics.sequence_scores_gpu(gtb, gen, j_only=metric == "j")      # the boundary measure only when it is asked for
    q = rows[:, 0 if metric == "j" else 2].copy()
    empty = (gtb.flatten(1).sum(1) == 0).cpu().numpy()
    mu = float(np.mean(q[~empty])) if (~empty).any() else float("nan")
    q[empty] = NO_OBJECT
    return mu, gen.to(torch.uint8), q


def _exhausted(q, frames, T):
    """not_avail_frames (interactions/eval.py:84-89): nothing left to annotate."""
    return not (set(range(T)) - set(np.where(q == NO_OBJECT)[0].tolist()) - set(frames))


def _upper_bound_frame(processor, gt, gt_thw, frames, metric):
    """get_frame_upper_bound (interactions/policies.py:90-117): try every remaining frame on a deep copy of the
    processor and keep the one with the best mean metric (last maximum wins, as the reference's ``>=``)."""
    best, best_f = -np.inf, -1
    for f in range(gt.shape[0]):
        if f in frames:
            continue
        p = copy.deepcopy(processor)
        p.interact(gt[f][None], f, download=False)
        mu, _, _ = frame_quality(p, gt_thw, frames + [f], metric)
        if mu >= best:
            best, best_f = mu, f
        del p
    return best_f


def run_policy(policy: str, processor, sample, rounds: int, metric: str = "j_and_f", qnet=None, rng=None):
    """One sample through ``rounds`` annotation rounds.  Returns dict(mu_metrics, annotation_times, frames,
    round_metrics): mu_metrics[r] / annotation_times[r] as the reference returns them; frames = annotated frames in
    order; round_metrics[r] = per-frame quality after round r."""
    assert policy in POLICIES, policy
    T = sample["num_frames"]
    dev = processor.prob.device
    gt = sample["gt"][0].to(dev)                                   # [T,1,H,W]
    gt_thw = gt[:, 0]
    images = sample["rgb"][0].to(dev) if policy == "qnet_mask" else None
    rng = rng or random
    # the evaluation of a round stays on the device (metrics.RoundScorer): one int per round crosses PCIe for the oracle policy, the
    # per-frame quality rows of the whole session are fetched once at the end
    scorer = metrics.RoundScorer(gt_thw, "j" if metric == "j" else "j_and_f", max_rounds=max(rounds, 1), no_object=NO_OBJECT)
    empty = scorer.empty_host
    valid = set(np.where(~empty)[0].tolist())
    frames, times, scored = [0], [MASK_SECONDS], 0
    propagated = 0                                                 # frames the engine really visited (rounds >= 2 only walk the spans next to the new annotation)
    for r in range(1, rounds + 1):
        if r >= T or (scored and not (valid - set(frames))):       # not_avail_frames (interactions/eval.py:84-89)
            continue
        f = frames[r - 1]
        processor.interact(gt[f][None], f, download=False)
        propagated += processor.stats()["frames"]
        worst, gen = scorer.score(processor, frames, keep_gen=policy == "qnet_mask")
        scored += 1
        if policy == "oracle_mask":
            sel = worst
        elif policy == "rand_mask":
            sel = rng.choice(sorted(set(range(T)) - set(frames)))
        elif policy == "qnet_mask":
            from .qnet import qnet_frame_selection
            sel = qnet_frame_selection(qnet, images, gen.float(), frames)
        else:
            sel = _upper_bound_frame(processor, gt, gt_thw, frames, metric)
        times.append(SKIP_SECONDS if empty[sel] else MASK_SECONDS)
        frames.append(sel)
    q = scorer.qualities()
    per_round = [q[i].copy() for i in range(scored)]
    mus = [float(np.mean(row[~empty])) if (~empty).any() else float("nan") for row in per_round]
    return dict(mu_metrics=mus, annotation_times=times[:-1], frames=frames, round_metrics=per_round, propagated_frames=propagated)


def run(root: str, imset: str, out_csv: str, prop_net, fuse_net, policy: str = "oracle_mask", rounds: int = 60,
        metric: str = "j_and_f", qnet=None, seed: int = 0, device: str = "cuda", lanes: int = 2, stats: dict = None):
    """Process this rank's share of the samples (`lanes` videos in flight); returns the gathered rows on every rank
    (rows: sample id, round, mu_metric, annotation_time, annotated frame, T, then T per-frame values, NaN-padded).
    `stats` (optional dict): this rank's `propagated_frames` (frames the engines visited) and `interactions` are added to it."""
    import torch.distributed as dist

    from mivos.inference_core import InferenceCore
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    ds = ClipDataset(root, imset)
    t_max = max(s[2] for s in ds.samples)
    mine = sorted(shard.lpt_assign([s[2] for s in ds.samples], world)[rank])
    width = 6 + t_max
    stats_lock = __import__("threading").Lock()

    def work(i, sample):
        rows = []
        t_c = time.perf_counter()
        proc = InferenceCore(prop_net, fuse_net, sample["rgb"], 1, engine_options=lane_engine_options(lanes))
        t_s = time.perf_counter()
        res = run_policy(policy, proc, sample, rounds, metric, qnet, random.Random(seed * 100003 + i))
        if stats is not None:
            with stats_lock:
                stats["create_s"] = stats.get("create_s", 0.0) + t_s - t_c
                stats["session_s"] = stats.get("session_s", 0.0) + time.perf_counter() - t_s
                stats["propagated_frames"] = stats.get("propagated_frames", 0) + res["propagated_frames"]
                stats["interactions"] = stats.get("interactions", 0) + len(res["mu_metrics"])
        for r, (mu, sec, q) in enumerate(zip(res["mu_metrics"], res["annotation_times"], res["round_metrics"])):
            row = np.full(width, np.nan, np.float32)
            row[:6] = (i, r, mu, sec, res["frames"][r], len(q))
            row[6:6 + len(q)] = q
            rows.append(row)
        return rows

    rows = run_lanes(root, imset, mine, lanes, work, device, stats)
    allrows = shard.gather_rows(np.stack(rows) if rows else np.zeros((0, width), np.float32), width)
    if rank == 0 and out_csv:
        os.makedirs(os.path.dirname(os.path.abspath(out_csv)), exist_ok=True)
        order = np.lexsort((allrows[:, 1], allrows[:, 0]))
        with open(out_csv, "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["video", "mu_metric", "annotation_time", "round"])
            for row in allrows[order]:
                wr.writerow([ds.name(int(row[0])), float(row[2]), int(row[3]), int(row[1])])
    return allrows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--imset", required=True)
    ap.add_argument("--policy", default="oracle_mask", choices=POLICIES)
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--lanes", type=int, default=2, help="videos in flight per GPU")
    ap.add_argument("--db", default="MOSE")
    ap.add_argument("--prop-weights", default="./model_weights/mivos/stcn.pth")
    ap.add_argument("--fusion-weights", default="./model_weights/mivos/fusion.pth")
    ap.add_argument("--qnet-weights", default="./model_weights/qnet/qnet.pth")
    ap.add_argument("--synthetic-weights", action="store_true", help="use the deterministic recipe (no checkpoints)")
    a = ap.parse_args()
    import torch.distributed as dist

    from . import synth
    from .params import FusionNet, PropagationNetwork
    torch.set_grad_enabled(False)
    shard.init_from_env()                                  # one process per GPU; RCCL unless STCN_DIST_BACKEND says otherwise
    prop, fuse, qnet = PropagationNetwork(), FusionNet(), None
    if a.policy == "qnet_mask":
        from .qnet import QualityNet
        qnet = QualityNet()
    if a.synthetic_weights:
        prop.load_state_dict(synth.recipe_state_dict(prop))
        fuse.load_state_dict(synth.recipe_state_dict(fuse))
        if qnet is not None:
            qnet.load_state_dict(synth.recipe_state_dict(qnet, seed=3))
    else:
        prop.load_state_dict(torch.load(a.prop_weights, map_location="cpu"))
        fuse.load_state_dict(torch.load(a.fusion_weights, map_location="cpu"))
        if qnet is not None:
            qnet.load_state_dict(torch.load(a.qnet_weights, map_location="cpu"))
    if qnet is not None:
        qnet = qnet.cuda().eval()
    out = os.path.join("Experiments", a.db, f"{a.policy}.csv")
    rows = run(a.root, a.imset, out, prop.eval(), fuse.eval(), a.policy, a.rounds, qnet=qnet, lanes=a.lanes)
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(f"{len(rows)} rounds -> {out}")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
