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
