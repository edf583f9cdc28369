"""Time the space-time memory read (pass 1 + threshold + pass 2 + merge/gather) with HIP events for growing banks and
both query shapes of the engine (one frame = 1620 queries; a 5-frame decode group = 8100).
Usage (GPU box): python tools/memread_bench.py [--k K]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import _lib  # noqa: E402

lib = _lib.lib()
k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 1
g = torch.Generator().manual_seed(0)
print(f"{'T':>4s} {'N':>7s} {'Q':>5s} {'k':>2s} {'ms':>8s} {'TFLOP/s':>8s} {'frac':>6s} {'GB/s alg':>9s}  plan(steps,ss,ns,nc1,spc1,nc2,spc2)")
cases = ((1, 1620), (5, 1620), (5, 8100), (14, 8100), (21, 8100), (52, 1620), (104, 1620))
if "--only" in sys.argv:                      # --only T,Q
    cases = (tuple(int(v) for v in sys.argv[sys.argv.index("--only") + 1].split(",")),)
for T, Q in cases:
    N = T * 1620
    mk = (torch.randn(N, 64, generator=g) * 0.8).cuda()
    qk = (torch.randn(Q, 64, generator=g) * 0.8).cuda()
    mv = torch.randn(k, N, 512, generator=g).cuda()
    ro = torch.empty(k, Q, 512, device="cuda")
    ms = C.c_float()
    plan = (C.c_int32 * 7)()
    _lib.check(lib.stcn_bench_memory_read(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(mk.data_ptr()),
                                          C.c_void_p(mv.data_ptr()), C.c_void_p(qk.data_ptr()), N, Q, k, 10,
                                          C.c_void_p(ro.data_ptr()), C.byref(ms), plan))
    fl = 2.0 * N * Q * 64
    by = 4.0 * (N * 65 + Q * 64 + k * Q * 50 * 512 + k * Q * 512)
    tf = fl / (ms.value * 1e-3) / 1e12
    print(f"{T:4d} {N:7d} {Q:5d} {k:2d} {ms.value:8.4f} {tf:8.1f} {tf / 157.3:6.3f} {by / (ms.value * 1e-3) / 1e9:9.0f}  {list(plan)}")
