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
