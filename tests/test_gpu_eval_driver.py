"""GPU: the config-5 mask-policy driver (eva_vos_amd.eval_driver) on the HIP engine against the literal CPU
restatement of the reference loops (oracle/policies_oracle.py driving the OracleCore), plus end-to-end runs."""
import csv
import random

import numpy as np
import pytest
import torch

from eva_vos_amd import eval_driver, fq_driver, synth
from oracle import policies_oracle as PO
from oracle import stcn_oracle as O

pytestmark = pytest.mark.gpu


def _sample(T, H, W):
    img = synth.synthetic_clip(T, H, W)
    gt = synth.synthetic_mask(T, H, W, 1)                         # [1,T,1,H,W]
    gt[0, T - 1] = 0                                              # one frame without the object (NO_OBJECT token)
    return {"rgb": img, "gt": gt, "num_frames": T, "name": "syn__1", "video": "syn"}


@pytest.mark.parametrize("metric", ["j", "j_and_f"])
def test_oracle_mask_policy_matches_the_cpu_restatement(nets, weights, metric):
    from mivos.inference_core import InferenceCore
    T, H, W, rounds = 7, 128, 160, 3
    s = _sample(T, H, W)
    proc = InferenceCore(nets[0], nets[1], s["rgb"].cuda(), 1)
    got = eval_driver.run_policy("oracle_mask", proc, s, rounds, metric)
    core = O.OracleCore(weights[0], weights[1], s["rgb"], 1)
    mus, times, frames = PO.mask_policy(core, s["gt"][0], rounds, lambda q, fr: int(np.argmin(q)), metric)
    assert got["frames"] == frames and got["annotation_times"] == times
    assert np.abs(np.array(got["mu_metrics"]) - np.array(mus)).max() <= 2e-3      # masks within 1e-3 IoU -> means too
    assert got["round_metrics"][0][T - 1] == eval_driver.NO_OBJECT
    assert got["round_metrics"][0][0] == 1.0                                       # annotated frame counts with its GT


def test_selectors_and_csv_end_to_end(nets, tmp_path):
    from eva_vos_amd.qnet import QualityNet
    imset = fq_driver.make_synthetic_tree(str(tmp_path / "db"), {"v0": (6, 112, 128, 2), "v1": (5, 112, 128, 1)})
    qnet = QualityNet()
    qnet.load_state_dict(synth.recipe_state_dict(qnet, seed=3))
    qnet = qnet.cuda().eval()
    rows = {}
    for policy in eval_driver.POLICIES:
        out = str(tmp_path / f"{policy}.csv")
        rows[policy] = eval_driver.run(str(tmp_path / "db"), imset, out, nets[0], nets[1], policy, rounds=3, qnet=qnet, seed=4)
        r = rows[policy]
        assert r.shape[1] == 6 + 6 and set(r[:, 0].astype(int)) == {0, 1, 2}
        for row in r:
            n = int(row[5])
            q = row[6:6 + n]
            assert np.all(((q >= 0) & (q <= 1)) | (q == eval_driver.NO_OBJECT)) and np.isnan(row[6 + n:]).all()
            assert 0 <= row[2] <= 1 and row[3] in (3, 80) and q[int(row[4])] == 1.0
        with open(out) as f:
            lines = list(csv.reader(f))
        assert lines[0] == ["video", "mu_metric", "annotation_time", "round"] and len(lines) == 1 + len(r)
        assert lines[1][0] == "v0__1" and lines[1][3] == "0"
    # every policy starts from frame 0: identical first round
    first = {p: r[r[:, 1] == 0][:, 2] for p, r in rows.items()}
    for p in eval_driver.POLICIES:
        assert np.array_equal(first[p], first["oracle_mask"])
    # the upper bound maximises the next round's mean by construction
    for sid in (0, 1, 2):
        pick = lambda p: rows[p][(rows[p][:, 0] == sid) & (rows[p][:, 1] == 1)][0, 2]    # noqa: E731
        assert pick("upper_bound_mask") >= max(pick("oracle_mask"), pick("rand_mask"), pick("qnet_mask")) - 1e-6
    # the random policy is reproducible under its seed
    again = eval_driver.run(str(tmp_path / "db"), imset, "", nets[0], nets[1], "rand_mask", rounds=3, seed=4)
    assert np.array_equal(np.nan_to_num(again), np.nan_to_num(rows["rand_mask"]))


def test_qnet_selection_on_device_equals_the_host_loop():
    from eva_vos_amd import qnet as Q
    net = Q.QualityNet()
    net.load_state_dict(synth.recipe_state_dict(net, seed=3))
    net = net.cuda().eval()
    T, H, W = 9, 120, 200
    frames = synth.synthetic_clip(T, H, W)[0].cuda()
    masks = synth.synthetic_mask(T, H, W, 1)[0, :, 0].cuda()
    imgs, m3 = Q.to_224(frames, masks)
    feats = net.extract_features(imgs, m3)
    for inter in ([0], [0, 4], [2, 3, 8]):
        assert Q.qnet_frame_selection(net, frames, masks, inter) == PO.farthest_frame(feats.cpu().numpy(), inter)


@pytest.mark.parametrize("metric,H,W", [("j", 120, 200), ("j_and_f", 120, 200), ("j_and_f", 101, 77)])
def test_round_scorer_on_the_device_equals_the_host_path_bit_for_bit(metric, H, W):
    """metrics.RoundScorer (stcn_metrics_round: compose + counts + fp64 quality + arg-min in one enqueue, one int per round over PCIe) against
    what the drivers did on the host until round 5 - eval_driver.frame_quality (sequence_scores_gpu + NumPy) and numpy.argmin: the per-frame
    values must be EQUAL as float64 bit patterns (same operations in the same order), the selection and the evaluated masks identical -
    with annotated frames, frames without the object, an all-empty prediction, ties (several annotated frames share J = 1) and a padded
    engine tensor (the scorer crops the engine's [T][nh][nw] masks itself)."""
    from eva_vos_amd import metrics

    class Proc:                                    # the attributes of an InferenceCore the scorer / frame_quality read
        pass
    T = 11
    rng = np.random.RandomState(H + W)
    gt = synth.synthetic_mask(T, H, W, 1, seed=5)[0, :, 0]
    gt[3] = 0                                                          # two frames without the object
    gt[9] = 0
    p = Proc()
    lh, lw = (-H) % 16 // 2, (-W) % 16 // 2
    p.nh, p.nw = H + (-H) % 16, W + (-W) % 16
    p.pad = (lw, p.nw - W - lw, lh, p.nh - H - lh)
    pred = synth.synthetic_mask(T, H, W, 1, seed=6)[0, :, 0].clone()
    pred[5] = 0                                                        # an empty prediction on a frame that has the object
    pred[3, :7, :9] = 1                                                # a prediction on a frame without the object
    noise = torch.from_numpy(rng.rand(T, H, W) < 0.02)
    pred = ((pred > 0.5) ^ noise).to(torch.uint8)
    masks = torch.zeros((T, 1, p.nh, p.nw), dtype=torch.uint8)
    masks[:, 0, lh:lh + H, lw:lw + W] = pred
    masks[:, 0, :lh] = 1                                               # garbage in the padding must not matter
    p.masks = masks.cuda()
    gt_dev = gt.cuda()
    sc = metrics.RoundScorer(gt_dev, metric, max_rounds=5, no_object=eval_driver.NO_OBJECT)
    assert sc.empty_host.tolist() == [t in (3, 9) for t in range(T)]
    # one new annotation per round, the last of the list (7 twice: a re-annotation); like the engine, a round only rewrites the masks of the
    # frames between the neighbouring annotated frames of the new one - the scorer recounts exactly those (incremental) - and the last
    # round flags every frame at once and is scored in full
    script = ([0], [0, 7], [0, 7, 2], [0, 2, 7, 7], list(range(T)))
    for r, annotated in enumerate(script):
        if 0 < r < 4:
            cur, others = annotated[-1], set(annotated[:-1]) - {annotated[-1]}
            lo, hi = max([f for f in others if f < cur] + [-1]), min([f for f in others if f > cur] + [T])
            flip = torch.from_numpy(rng.rand(hi - lo - 1, H, W) < 0.01).to(torch.uint8).cuda()
            p.masks[lo + 1:hi, 0, lh:lh + H, lw:lw + W] ^= flip           # what a propagation round may change
        sel, gen = sc.score(p, annotated, incremental=r < 4)
        mu, gen_ref, q_ref = eval_driver.frame_quality(p, gt_dev, sorted(set(annotated)), metric)
        q = sc.qualities()[r]
        assert q.dtype == np.float64 and np.array_equal(q.view(np.uint64), q_ref.view(np.uint64)), (r, np.abs(q - q_ref).max())
        assert sel == int(np.argmin(q_ref)), (r, sel, int(np.argmin(q_ref)))
        assert torch.equal(gen, gen_ref), r
        assert all(q[f] == (eval_driver.NO_OBJECT if f in (3, 9) else 1.0) for f in annotated)
    assert sc.qualities().shape == (5, T)
