"""Padding helpers with the reference's semantics (mivos/tensor_util.py:62-94), used by callers of
InferenceCore (e.g. interactions/eval.py unpads ``processor.prob`` with ``processor.pad``)."""
import torch.nn.functional as F


def pad_divide_by(in_img, d, in_size=None):
    h, w = in_img.shape[-2:] if in_size is None else in_size
    dh, dw = (-h) % d, (-w) % d
    lh, lw = dh // 2, dw // 2
    pad = (lw, dw - lw, lh, dh - lh)
    return F.pad(in_img, pad), pad


def unpad(img, pad):
    lw, uw, lh, uh = pad
    H, W = img.shape[-2:]
    return img[..., lh:H - uh, lw:W - uw]
