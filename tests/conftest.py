import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

torch.set_grad_enabled(False)
# the CPU oracle thrashes with hundreds of intra-op threads on the GPU box's host
torch.set_num_threads(min(32, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def weights():
    """(prop_state_dict, fuse_state_dict) from the frozen synthetic recipe (seed 0)."""
    from eva_vos_amd import synth
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    return synth.recipe_state_dict(PropagationNetwork()), synth.recipe_state_dict(FusionNet())


@pytest.fixture(scope="session")
def nets(weights):
    """Parameter containers loaded with the recipe (what a user would pass to InferenceCore)."""
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    p, f = PropagationNetwork(), FusionNet()
    p.load_state_dict(weights[0], strict=True)
    f.load_state_dict(weights[1], strict=True)
    return p.eval(), f.eval()


@pytest.fixture(scope="session")
def weights_multi():
    """The MULTI-OBJECT recipe (seed 2: Philox draws + the fitted last layers, eva_vos_amd/synth.py RECIPES): the decoder separates
    several objects, so multi-object mask parity can be stated on (almost) all pixels."""
    from eva_vos_amd import synth
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    return synth.recipe_state_dict(PropagationNetwork(), 2), synth.recipe_state_dict(FusionNet(), 2)


@pytest.fixture(scope="session")
def nets_multi(weights_multi):
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    p, f = PropagationNetwork(), FusionNet()
    p.load_state_dict(weights_multi[0], strict=True)
    f.load_state_dict(weights_multi[1], strict=True)
    return p.eval(), f.eval()


def load_golden(tag):
    return dict(np.load(os.path.join(GOLD, f"{tag}.npz")))


def sample_of(t, stride):
    return t.detach().float().cpu().numpy().reshape(-1)[::stride]


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def iou(a, b):
    a, b = np.asarray(a).astype(bool), np.asarray(b).astype(bool)
    u = (a | b).sum()
    return 1.0 if u == 0 else float((a & b).sum() / u)


def frame_miss(a, b, min_union=64):
    """Worst PER-FRAME (1 - IoU) of two boolean [T,H,W] masks over the frames whose union has >= min_union pixels (a volume IoU
    over the clip hides one bad frame among many), and the frame it occurs on."""
    a, b = np.asarray(a).astype(bool).reshape(len(a), -1), np.asarray(b).astype(bool).reshape(len(b), -1)
    u, n = (a | b).sum(1), (a & b).sum(1)
    ok = u >= min_union
    if not ok.any():
        return 0.0, -1
    miss = np.where(ok, 1.0 - n / np.maximum(u, 1), 0.0)
    return float(miss.max()), int(miss.argmax())


# Allowance over the reference's own spread (round 6: 1.5, was 3 - a 3 x allowance would hide a 3 x regression; measured HIP-vs-oracle
# differences sit at ~1.0 x the reference's thread-count noise, so 1.5 x is margin for which near-ties happen to flip, not for drift)
NOISE_X = 1.5


def frame_bound(noise_col4, union_px):
    """Per-frame mask bound: the north_star 1e-3, or NOISE_X (1.5) x the reference's own worst per-frame difference between its 1/2/4/8-
    thread runs on that fixture (tests/golden/selfnoise.npz, column 4), or - small objects - two pixels, whichever is larger."""
    return max(1e-3, NOISE_X * float(noise_col4), 2.0 / max(float(union_px), 1.0))


def clip_bound(noise_col0=0.0):
    """Per-object mask bound on a whole clip (1 - IoU): the north_star 1e-3; only where the reference's OWN clip-level spread on the
    nearest fixture exceeds it / NOISE_X (small frames with k > 1: 1.16e-3 on seqC / seqD, saturated aggregation) NOISE_X x that."""
    return max(1e-3, NOISE_X * float(noise_col0))
