"""CPU: QualityNet container vs the reference golden, frame selection vs the literal loop restatement."""
import os

import numpy as np
import torch

from eva_vos_amd import qnet as Q
from eva_vos_amd import synth
from oracle import policies_oracle as PO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "qnet.npz")


def _inputs(n=3, seed=11):          # same streams as oracle/gen_golden_qnet.py
    g = np.random.Generator(np.random.Philox(key=[seed, 224]))
    rgb = torch.from_numpy(g.normal(0, 1, (n, 3, 224, 224)).astype(np.float32))
    msk = torch.from_numpy((g.uniform(0, 1, (n, 1, 224, 224)) > 0.6).astype(np.float32)).expand(-1, 3, -1, -1)
    return rgb, msk.contiguous()


def test_qualitynet_state_dict_and_features_match_the_reference_golden():
    gold = np.load(GOLD)
    net = Q.QualityNet().eval()
    sd = net.state_dict()
    assert list(sd) == list(gold["names"])                       # qnet.pth loads strictly
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(gold["shapes"])
    net.load_state_dict(synth.recipe_state_dict(net, seed=3))
    with torch.no_grad():
        rgb, msk = _inputs()
        feats, logits = net.extract_features(rgb, msk), net(rgb, msk)
    scale = np.abs(gold["features"]).max()
    assert np.abs(feats.numpy() - gold["features"]).max() <= 1e-5 * scale      # fp32, same ops: thread-count noise only
    assert np.abs(logits.numpy() - gold["logits"]).max() <= 1e-5 * np.abs(gold["logits"]).max()


def test_frame_selection_equals_the_reference_loop():
    g = np.random.default_rng(5)
    for T, n_int in ((7, 1), (30, 4), (64, 9)):
        feats = g.normal(0, 3, (T, 1024)).astype(np.float32)
        inter = sorted(g.choice(T, n_int, replace=False).tolist())
        assert Q.select_farthest(torch.from_numpy(feats), inter) == PO.farthest_frame(feats, inter)
    # exact ties (duplicated feature rows): the first frame with the largest min-distance wins, as the strict '>' scan
    feats = g.normal(0, 1, (6, 16)).astype(np.float32)
    feats[4] = feats[2]
    feats[0] = 0
    feats[2] = feats[4] = 10.0
    assert Q.select_farthest(torch.from_numpy(feats), [0]) == PO.farthest_frame(feats, [0]) == 2
    # only interacted frames left at distance zero: index 0 like the reference
    feats = np.zeros((4, 8), np.float32)
    assert Q.select_farthest(torch.from_numpy(feats), [1, 2]) == PO.farthest_frame(feats, [1, 2]) == 0


def test_resize_to_224_semantics():
    g = np.random.default_rng(2)
    frames = torch.from_numpy(g.normal(0, 1, (2, 3, 100, 150)).astype(np.float32))
    masks = torch.from_numpy((g.uniform(0, 1, (2, 100, 150)) > 0.5).astype(np.float32))
    imgs, m3 = Q.to_224(frames, masks)
    assert imgs.shape == (2, 3, 224, 224) and m3.shape == (2, 3, 224, 224)
    iy = np.floor(np.arange(224) * 100 / 224).astype(int)        # nearest: source index floor(i * in / out)
    ix = np.floor(np.arange(224) * 150 / 224).astype(int)
    assert np.array_equal(m3[:, 0].numpy(), masks.numpy()[:, iy][:, :, ix])
    assert torch.equal(m3[:, 0], m3[:, 1]) and torch.equal(m3[:, 0], m3[:, 2])
    # antialiased bicubic of a constant image is that constant
    c, _ = Q.to_224(torch.full((1, 3, 100, 150), 0.7), masks[:1])
    assert torch.allclose(c, torch.full_like(c, 0.7), atol=1e-5)
