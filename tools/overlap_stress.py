"""Overlap stress matrix: does a conv kernel on another stream perturb a concurrently running victim kernel?
Usage (GPU box): python tools/overlap_stress.py [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import _lib  # noqa: E402

lib = _lib.lib()
torch.cuda.init()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FLAGS = {}
modes = [(-1, "no conv"), (0, "fp32 conv"), (1, "f16x3 conv")]
modes += [(1 | (f << 4), "f16x3 " + n) for f, n in FLAGS.items()]
for victim, vname in ((1, "gather_sum (float2 fma)"), (2, "gather_sum (v_fmac)")):
    for mode, mname in modes:
        bad, first = C.c_int(), C.c_int()
        _lib.check(lib.stcn_debug_overlap(victim, mode, iters, C.byref(bad), C.byref(first)))
        q, ch = (first.value // 512, first.value % 512) if first.value >= 0 else (-1, -1)
        print(f"victim {vname:24s} beside {mname:28s}: {bad.value:3d} / {iters} outputs differ (first at query {q}, channel {ch})")
