"""Helpers for the -m gpu tests: device tensors in the engine's NHWC layout, C-ABI calls."""
import ctypes as C

import torch

from eva_vos_amd import _lib
from eva_vos_amd.inference_core import _model_for


def dev(t):
    if t.dtype in (torch.int32, torch.uint8):
        return t.detach().to("cuda").contiguous()
    return t.detach().to("cuda", torch.float32).contiguous()


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def nhwc(x):           # [B,C,H,W] -> [B,H,W,C] contiguous on device
    return dev(x.permute(0, 2, 3, 1))


def rows_to_nchw(x, h, w):   # [B,h*w,C] (device) -> [B,C,h,w] cpu
    return x.reshape(x.shape[0], h, w, -1).permute(0, 3, 1, 2).contiguous().cpu()


def model_handle(nets):
    return _model_for(nets[0], nets[1], torch.cuda.current_device()).handle


def call(name, *args):
    """Call a C-ABI entry point.  Tensor arguments are passed as tensors (kept alive for the whole call
    and converted to device pointers here); None -> NULL."""
    keep = [dev(a) if isinstance(a, torch.Tensor) and (a.device.type != "cuda" or not a.is_contiguous()) else a
            for a in args]
    conv = [C.c_void_p(a.data_ptr()) if isinstance(a, torch.Tensor) else a for a in keep]
    _lib.check(getattr(_lib.lib(), name)(*conv), name)
    torch.cuda.synchronize()
    del keep
