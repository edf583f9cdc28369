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


class RoundScorer:
    """The per-round evaluation of an annotation session, kept on the device (``stcn_metrics_round``): after every ``interact()`` the
    reference's loops compute per-frame J or J&F of the propagated masks against the ground truth (annotated frames counting with their
    ground truth, the NO_OBJECT token for frames without the object: interactions/eval.py:27-81) and the oracle policy takes the arg-min
    (interactions/mask.py:130-133).  Here one C call enqueues compose + counts + fp64 quality + arg-min on the engine's stream; only the
    selected frame (4 bytes) crosses PCIe per round - the host waits for it on a BLOCKING event (it sleeps instead of spinning a core) -
    and the quality rows of all rounds are fetched together at the end of the session (``qualities()``).  Bit-identical to the host path
    (``sequence_scores_gpu`` + NumPy), which the tests assert."""

    def __init__(self, gt_thw, metric: str = "j", max_rounds: int = 64, no_object: float = 20.0):
        import torch
        assert gt_thw.is_cuda and gt_thw.dim() == 3
        self.metric, self.no_object = metric, float(no_object)
        self.dev = gt_thw.device
        self.gt = (gt_thw > 0.5 if gt_thw.is_floating_point() else gt_thw != 0).to(torch.uint8).contiguous()
        self.T, self.H, self.W = (int(v) for v in self.gt.shape)
        empty = self.gt.flatten(1).sum(1) == 0
        self.noobj = empty.to(torch.uint8).contiguous()
        self.empty_host = empty.cpu().numpy()                         # ONE sync per sample: which frames carry the NO_OBJECT token
        self.annotated = torch.zeros(self.T, dtype=torch.uint8, device=self.dev)
        self.flags_host = torch.zeros(self.T, dtype=torch.uint8).pin_memory()
        self.counts = torch.empty((self.T, 6), dtype=torch.int32, device=self.dev)
        self.scratch = None if metric == "j" else torch.empty((self.T * self.H * self.W,), dtype=torch.uint8, device=self.dev)
        self.quality = torch.empty((max_rounds, self.T), dtype=torch.float64, device=self.dev)
        self.select = torch.empty((max_rounds,), dtype=torch.int32, device=self.dev)
        self.select_host = torch.empty((max_rounds,), dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event(blocking=True)
        self.rounds = 0

    def score(self, processor, annotated_frames, keep_gen: bool = True, incremental: bool = True):
        """Enqueue the evaluation of the round just propagated by ``processor`` and return (selected frame, gen): gen = uint8 [T,H,W] on the
        device (the evaluated masks; a fresh tensor when keep_gen, else a scratch that the next round overwrites).  ``annotated_frames``:
        every frame annotated so far, the one annotated in THIS round last.  ``incremental``: from the second round on only the frames the
        round can have changed - between the neighbouring annotated frames of the new one - are composed and counted again (the masks of the
        others are what they were: the engine only rewrites the probabilities of the frames it visits)."""
        import ctypes as C

        import torch

        from . import _lib
        r = self.rounds
        if r >= self.quality.shape[0]:
            raise RuntimeError("RoundScorer: more rounds than max_rounds")
        lw, uw, lh, uh = processor.pad
        frames = [int(f) for f in annotated_frames]
        cur, others = frames[-1], set(frames[:-1]) - {frames[-1]}
        t0, t1 = 0, self.T
        if incremental and r > 0:
            t0 = max([f for f in others if f < cur] + [-1]) + 1
            t1 = min([f for f in others if f > cur] + [self.T])
        with torch.cuda.device(self.dev):
            self.flags_host.zero_()                                   # (the previous round's copy is done: every round ends in a wait)
            self.flags_host[sorted(set(frames))] = 1
            self.annotated.copy_(self.flags_host, non_blocking=True)  # T bytes H2D from pinned memory
            prev = getattr(self, "_gen", None)
            if prev is None:
                gen = torch.empty((self.T, self.H, self.W), dtype=torch.uint8, device=self.dev)
                t0, t1 = 0, self.T
            elif keep_gen:
                gen = prev.clone() if (t0, t1) != (0, self.T) else torch.empty_like(prev)      # the frames outside [t0, t1) carry over
            else:
                gen = prev
            self._gen = gen
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
            _lib.check(_lib.lib().stcn_metrics_round(
                C.c_void_p(torch.cuda.current_stream().cuda_stream), p(processor.masks), processor.nh, processor.nw, lh, lw, p(self.gt),
                p(self.annotated), p(self.noobj), self.T, self.H, self.W, t0, t1, 1 if self.metric == "j" else 0, self.no_object, p(gen),
                p(self.scratch), p(self.counts), p(self.quality[r]), p(self.select[r:r + 1])), "stcn_metrics_round")
            self.select_host[r:r + 1].copy_(self.select[r:r + 1], non_blocking=True)
            self.event.record()
            self.event.synchronize()                                  # blocking wait: the lane's host thread sleeps until the round is done
        self.rounds = r + 1
        return int(self.select_host[r]), gen

    def qualities(self):
        """float64 [rounds, T]: the per-frame quality of every round scored so far (one D2H copy)."""
        return self.quality[: self.rounds].cpu().numpy()
