"""Time the space-time memory read (affinity_topk + merge_readout) for growing banks.
Usage (GPU box): python tools/memread_bench.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import _lib  # noqa: E402

lib = _lib.lib()
Q = 1620
g = torch.Generator().manual_seed(0)
for T in (1, 2, 5, 14, 21):
    N = T * Q
    mk = (torch.randn(N, 64, generator=g) * 0.8).cuda()
    qk = (torch.randn(Q, 64, generator=g) * 0.8).cuda()
    mv = torch.randn(1, N, 512, generator=g).cuda()
    ro = torch.empty(1, Q, 512, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (s, C.c_void_p(mk.data_ptr()), C.c_void_p(mv.data_ptr()), C.c_void_p(qk.data_ptr()), N, Q, 1, None, None,
            C.c_void_p(ro.data_ptr()))
    for _ in range(2):
        _lib.check(lib.stcn_test_memory_read(*args))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        _lib.check(lib.stcn_test_memory_read(*args))
    torch.cuda.synchronize()
    print(f"T={T:3d} N={N:6d}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call (includes hook malloc/sync overhead)")
