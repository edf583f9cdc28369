"""CPU restatement of the reference's mask-annotation policy loops - TEST INFRASTRUCTURE ONLY (see oracle/stcn_oracle.py).

Follows the reference literally (host loops, NumPy): the checker for ``eva_vos_amd.eval_driver`` / ``eva_vos_amd.qnet``.
Parity: the reference's own ``interactions/`` package cannot be imported offline (cv2, skimage, torchmetrics,
torchvision.transforms are absent - SURVEY.md section 8(c)), so these loops are pinned by reading, not by execution:
"parity unpinned" for this file; the propagation they drive (OracleCore) is pinned by tests/golden.
"""
import numpy as np

from eva_vos_amd import metrics

NO_OBJECT = 20          # interactions/eval.py:67


def min_l2(interacted_features, curr):
    """get_min_l2_dist (interactions/policies.py:21-35)."""
    best = np.inf
    for f in interacted_features:
        d = np.linalg.norm(curr - f)
        if d < best:
            best = d
    return best


def farthest_frame(features, interacted):
    """Selection loop shared by qnet_frame_selection / get_frame_l2 (interactions/policies.py:48-60,76-88)."""
    features = np.asarray(features)
    inter = features[list(interacted)]
    best, frame = -np.inf, None
    for i in range(len(features)):
        d = min_l2(inter, features[i])
        if d > best:
            best, frame = d, i
    return frame


def frame_quality(core, gt_thw, interacted, metric="j_and_f"):
    """eval_processor_metric for mask annotations (interactions/eval.py:27-81) on an OracleCore."""
    lw, uw, lh, uh = core.pad
    prob = core.prob[:, :, 0, lh:core.prob.shape[-2] - uh, lw:core.prob.shape[-1] - uw]
    seg = prob.argmax(0).numpy().astype(bool)                      # get_segmentations (:8-24)
    gt = np.asarray(gt_thw) > 0.5
    q_obj, q_all = [], []
    for t in range(len(gt)):
        pred = gt[t] if t in interacted else seg[t]
        if gt[t].sum() == 0:
            q_all.append(NO_OBJECT)
            continue
        j = metrics.jaccard(gt[t], pred)
        v = j if metric == "j" else 0.5 * j + 0.5 * metrics.f_measure(gt[t], pred)
        q_obj.append(v)
        q_all.append(v)
    return float(np.mean(q_obj)), np.array(q_all, np.float64)


def mask_policy(core, gt, rounds, select, metric="j_and_f"):
    """Skeleton shared by qnet_mask / rand_mask / oracle_mask / upper_bound_mask (interactions/mask.py:10-103,196-227).
    gt: float tensor [T,1,H,W]; select(q, frames) -> next frame.  Returns (mu_metrics, annotation_times, frames)."""
    T = gt.shape[0]
    frames, q = [0], None
    mus, times = [], [80]
    for r in range(1, rounds + 1):
        if r >= T:
            continue
        if q is not None and not (set(range(T)) - set(np.where(q == NO_OBJECT)[0].tolist()) - set(frames)):
            continue
        f = frames[r - 1]
        core.interact(gt[f][None].clone(), f)
        mu, q = frame_quality(core, gt[:, 0].numpy(), frames, metric)
        mus.append(mu)
        sel = select(q, frames)
        times.append(3 if q[sel] == NO_OBJECT else 80)
        frames.append(sel)
    return mus, times[:-1], frames
