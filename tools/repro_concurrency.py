"""Does a conv kernel running on another stream perturb a concurrently running memory read?
Usage (GPU box): STCN_PRECISION=f16x3|f32 python tools/repro_concurrency.py"""
import ctypes as C
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import _lib  # noqa: E402

lib = _lib.lib()
torch.cuda.init()
g = torch.Generator().manual_seed(0)
N, Q = 1620, 1620
mk = (torch.randn(N, 64, generator=g) * 0.8).cuda()
qk = (torch.randn(Q, 64, generator=g) * 0.8).cuda()
mv = torch.randn(1, N, 512, generator=g).cuda()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
stop = False


def conv_loop():
    B, H, W, Cin, Cout, K = 1, 60, 108, 256, 256, 3
    x = torch.randn(B, H, W, Cin, generator=torch.Generator().manual_seed(1)).cuda()
    w = (torch.randn(Cout, K, K, Cin, generator=torch.Generator().manual_seed(2)) * 0.02).cuda()
    b = torch.zeros(Cout).cuda()
    y = torch.empty(B, H, W, Cout, device="cuda")
    n = 0
    with torch.cuda.stream(sB):
        while not stop:
            _lib.check(lib.stcn_test_conv(C.c_void_p(sB.cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()),
                                          C.c_void_p(b.data_ptr()), None, C.c_void_p(y.data_ptr()), B, H, W, Cin, Cout, K, K, 1, 1, 0, 0))
            n += 1
    print("conv launches:", n)


def read_once():
    ro = torch.empty(1, Q, 512, device="cuda")
    _lib.check(lib.stcn_test_memory_read(C.c_void_p(sA.cuda_stream), C.c_void_p(mk.data_ptr()), C.c_void_p(mv.data_ptr()),
                                         C.c_void_p(qk.data_ptr()), N, Q, 1, None, None, C.c_void_p(ro.data_ptr())))
    return ro


with torch.cuda.stream(sA):
    ref = read_once().clone()
    solo_bad = sum(int(not torch.equal(read_once(), ref)) for _ in range(20))
print("solo: mismatching reads", solo_bad, "of 20")
t = threading.Thread(target=conv_loop)
t.start()
time.sleep(0.2)
bad, worst = 0, 0.0
with torch.cuda.stream(sA):
    for i in range(200):
        r = read_once()
        if not torch.equal(r, ref):
            bad += 1
            d = (r - ref).abs()
            worst = max(worst, float(d.max()))
            if bad <= 3:
                idx = torch.nonzero(d[0] > 0)
                print("  mismatch: elements", idx.shape[0], "first (q,ch):", idx[:6].tolist())
stop = True
t.join()
print(f"concurrent ({os.environ.get('STCN_PRECISION', 'f32')} conv on the other stream): mismatching reads {bad} of 200, worst |diff| {worst:.4g}")
