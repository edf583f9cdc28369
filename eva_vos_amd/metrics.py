"""J (region IoU), F (boundary F-measure) and J&F for binary masks - own NumPy/SciPy implementation of
the measures the reference's callers compute per frame after every interaction
(interactions/metrics.py:24-34 ``get_j_and_f``, :38-97 boundary map, :100-160 ``f_measure``;
interactions/eval.py:27-81).  Caller-side code, integer-exact."""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def jaccard(gt: np.ndarray, pred: np.ndarray) -> float:
    gt, pred = np.asarray(gt, bool), np.asarray(pred, bool)
    union = (gt | pred).sum()
    return 0.0 if union == 0 else float((gt & pred).sum() / union)


def boundary_map(seg: np.ndarray) -> np.ndarray:
    """1-pixel boundary, offset half a pixel towards the origin (the classic seg2bmap at equal size)."""
    seg = np.asarray(seg, bool)
    e, s, se = np.zeros_like(seg), np.zeros_like(seg), np.zeros_like(seg)
    e[:, :-1], s[:-1, :], se[:-1, :-1] = seg[:, 1:], seg[1:, :], seg[1:, 1:]
    b = (seg ^ e) | (seg ^ s) | (seg ^ se)
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = False
    return b


def _disk(r: int) -> np.ndarray:
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r


def f_measure(gt: np.ndarray, pred: np.ndarray, bound_th: float = 0.008) -> float:
    gt, pred = np.asarray(gt, bool), np.asarray(pred, bool)
    r = int(bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(gt.shape)))
    fb, gb = boundary_map(pred), boundary_map(gt)
    se = _disk(r)
    n_fg, n_gt = int(fb.sum()), int(gb.sum())
    if n_fg == 0 and n_gt > 0:
        p, rc = 1.0, 0.0
    elif n_fg > 0 and n_gt == 0:
        p, rc = 0.0, 1.0
    elif n_fg == 0 and n_gt == 0:
        p, rc = 1.0, 1.0
    else:
        p = float((fb & ndimage.binary_dilation(gb, se)).sum() / n_fg)
        rc = float((gb & ndimage.binary_dilation(fb, se)).sum() / n_gt)
    return 0.0 if p + rc == 0 else 2 * p * rc / (p + rc)


def j_and_f(gt: np.ndarray, pred: np.ndarray) -> float:
    return 0.5 * jaccard(gt, pred) + 0.5 * f_measure(gt, pred)


def sequence_scores(gt: np.ndarray, pred: np.ndarray, every: int = 1) -> np.ndarray:
    """[T,H,W] masks -> rows (frame, J, F, J&F) for frames 0, every, 2*every, ..."""
    rows = []
    for t in range(0, gt.shape[0], every):
        j, f = jaccard(gt[t], pred[t]), f_measure(gt[t], pred[t])
        rows.append((t, j, f, 0.5 * (j + f)))
    return np.asarray(rows, np.float32)


# ------------------------------------------------------------------------------------------------ GPU path
def _scores_from_counts(c: np.ndarray) -> np.ndarray:
    """[T,6] integer counts -> rows (J, F, J&F) with the reference's special cases (metrics.py:141-158)."""
    out = np.zeros((c.shape[0], 3), np.float64)
    for t, (inter, union, n_gt, n_fg, gt_m, fg_m) in enumerate(c.tolist()):
        j = 0.0 if union == 0 else inter / union
        if n_fg == 0 and n_gt > 0:
            p, r = 1.0, 0.0
        elif n_fg > 0 and n_gt == 0:
            p, r = 0.0, 1.0
        elif n_fg == 0 and n_gt == 0:
            p, r = 1.0, 1.0
        else:
            p, r = fg_m / n_fg, gt_m / n_gt
        f = 0.0 if p + r == 0 else 2 * p * r / (p + r)
        out[t] = (j, f, 0.5 * (j + f))
    return out


def sequence_scores_gpu(gt, pred, j_only: bool = False):
    """J, F, J&F per frame on the GPU (HIP kernels behind ``stcn_metrics_jf_counts``).
    gt, pred: torch uint8/bool tensors [T,H,W] on the same cuda device (non-zero = object).
    Returns float64 [T,3]; only the 6*T integer counts cross PCIe.  ``j_only``: the region measure alone (``stcn_metrics_j_counts``:
    no boundary maps, no disk matching) - column 0 is J, columns 1 and 2 are NaN."""
    import ctypes as C

    import torch

    from . import _lib
    gt = (gt != 0).to(torch.uint8).contiguous()
    pred = (pred != 0).to(torch.uint8).contiguous()
    assert gt.is_cuda and pred.is_cuda and gt.shape == pred.shape and gt.dim() == 3
    T, H, W = gt.shape
    with torch.cuda.device(gt.device):
        counts = torch.empty((T, 6), dtype=torch.int32, device=gt.device)
        if j_only:
            _lib.check(_lib.lib().stcn_metrics_j_counts(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(gt.data_ptr()),
                                                        C.c_void_p(pred.data_ptr()), T, H, W, C.c_void_p(counts.data_ptr())), "stcn_metrics_j_counts")
            c = counts.cpu().numpy()
            out = np.full((T, 3), np.nan)
            out[:, 0] = np.where(c[:, 1] > 0, c[:, 0] / np.maximum(c[:, 1], 1), 0.0)
            return out
        scratch = torch.empty((T * H * W,), dtype=torch.uint8, device=gt.device)
        _lib.check(_lib.lib().stcn_metrics_jf_counts(C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                     C.c_void_p(gt.data_ptr()), C.c_void_p(pred.data_ptr()), T, H, W,
                                                     C.c_void_p(counts.data_ptr()), C.c_void_p(scratch.data_ptr())),
                   "stcn_metrics_jf_counts")
        c = counts.cpu().numpy()
    return _scores_from_counts(c)
