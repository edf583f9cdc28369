"""J / F against fixtures captured from the reference's interactions/metrics.py (oracle/gen_golden_metrics.py).
Boundary maps are the reference's own ``_seg2bmap`` output (pure NumPy there); F and J&F went through stand-ins for
cv2.dilate / skimage disk / torchmetrics JaccardIndex (not installed in the build container) and are named ``*_standin``."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from eva_vos_amd import metrics

CASES = ("small", "odd", "p480")


def unpack(g, tag):
    T, H, W = [int(v) for v in g[f"{tag}.shape"]]
    u = lambda name: np.unpackbits(g[f"{tag}.{name}"])[: T * H * W].reshape(T, H, W).astype(bool)   # noqa: E731
    return u("gt"), u("pred"), u("bmap_gt"), u("bmap_pred")


@pytest.mark.parametrize("tag", CASES)
def test_cpu_metrics_match_the_reference_fixture(tag):
    g = load_golden("metrics")
    gt, pr, bg, bp = unpack(g, tag)
    assert gt[2, 0].all() and gt[2, -1, -1] and pr[2, :, 0].all(), "fixture must hold objects touching the image edges"
    for t in range(gt.shape[0]):
        assert np.array_equal(metrics.boundary_map(gt[t]), bg[t]), (tag, t)            # reference _seg2bmap, bit for bit
        assert np.array_equal(metrics.boundary_map(pr[t]), bp[t]), (tag, t)
        f = metrics.f_measure(gt[t], pr[t])
        assert abs(f - g[f"{tag}.f_standin"][t]) < 1e-12, (tag, t, f)
        jf = metrics.j_and_f(gt[t], pr[t])
        assert abs(jf - g[f"{tag}.jf_standin"][t]) < 1e-6, (tag, t, jf)                 # reference J is a float32 tensor op


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_gpu_metrics_match_the_reference_fixture(tag):
    """The HIP J/F kernel (stcn_metrics_jf_counts) directly against the reference-derived values."""
    g = load_golden("metrics")
    gt, pr, bg, bp = unpack(g, tag)
    got = metrics.sequence_scores_gpu(torch.from_numpy(gt).cuda(), torch.from_numpy(pr).cuda())
    assert np.abs(got[:, 1] - g[f"{tag}.f_standin"]).max() < 1e-12
    assert np.abs(got[:, 2] - g[f"{tag}.jf_standin"]).max() < 1e-6
    # the kernel's boundary-pixel counts equal the reference boundary maps' pixel counts
    import ctypes as C
    from eva_vos_amd import _lib
    T, H, W = gt.shape
    a, b = torch.from_numpy(gt).cuda().to(torch.uint8), torch.from_numpy(pr).cuda().to(torch.uint8)
    counts = torch.empty((T, 6), dtype=torch.int32, device="cuda")
    scratch = torch.empty((T * H * W,), dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().stcn_metrics_jf_counts(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(a.data_ptr()),
                                                 C.c_void_p(b.data_ptr()), T, H, W, C.c_void_p(counts.data_ptr()),
                                                 C.c_void_p(scratch.data_ptr())))
    c = counts.cpu().numpy()
    assert c[:, 2].tolist() == bg.reshape(T, -1).sum(1).tolist() and c[:, 3].tolist() == bp.reshape(T, -1).sum(1).tolist()
