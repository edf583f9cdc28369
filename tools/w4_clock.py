"""GPU box, DIAGNOSTIC builds only: the clock the chip holds INSIDE the main loop of wino4_gemm_kernel (MI355X_MICROARCH.md, DVFS give-back (6):
delta s_memtime / delta s_memrealtime x 100 MHz, stamped by wave 0 of every workgroup, median over workgroups, after >= 2 s of launches).
  STCN_LIB=eva_vos_amd/csrc/build/exp/libstcn_hip_clk1.so python tools/w4_clock.py [--gap-ms G] [--seconds S] [--shape B,H,W,Cin,Cout]
    clk1 = the shipped loop + stamps; clk2 = every MFMA replaced by 16 v_fma_f32 on its accumulator (same issue cycles and loads, no matrix math).
  --gap-ms G: an idle gap of G ms after every launch (host sync + sleep): does the clock recover when the chip rests?"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import _lib  # noqa: E402


def arg(name, dflt):
    return type(dflt)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt


gap_ms, seconds = arg("--gap-ms", 0.0), arg("--seconds", 2.5)
B, H, W, Cin, Cout = (int(v) for v in arg("--shape", "5,120,216,256,256").split(","))
os.environ["STCN_BENCH_CONV_F4"] = "1"
lib = _lib.lib()
dbg = getattr(lib, "stcn_debug_w4_clock", None)
assert dbg is not None, "not a diagnostic build: make EXTRA=-DSTCN_W4_CLOCK=1 OBJDIR=build/objclk1 OUT=build/exp/libstcn_hip_clk1.so, then STCN_LIB=..."
dbg.restype, dbg.argtypes = C.c_int, [C.c_void_p, C.c_int]
torch.cuda.init()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ms, fl = C.c_float(), C.c_double()


def run(iters):
    _lib.check(lib.stcn_bench_conv(s, B, H, W, Cin, Cout, 3, 3, 1, 1, 0, iters, C.byref(ms), C.byref(fl)))
    return ms.value


per = run(10)                                                   # ms per conv (transform + GEMM)
t0, launches, times = time.perf_counter(), 0, []
if gap_ms > 0:
    while time.perf_counter() - t0 < seconds:
        times.append(run(1))
        launches += 1
        torch.cuda.synchronize()
        time.sleep(gap_ms * 1e-3)
else:
    n = max(50, int(seconds * 1e3 / per))
    times.append(run(n))
    launches = n
tiles = B * ((H + 3) // 4) * ((W + 3) // 4)
n_wg = min(8192, ((tiles + 63) // 64) * (Cout // 32))
buf = (C.c_ulonglong * (2 * n_wg))()
assert dbg(buf, n_wg) == 0
a = np.array(buf[:], np.float64).reshape(-1, 2)
a = a[(a[:, 0] > 0) & (a[:, 1] > 0)]
clk = a[:, 0] / a[:, 1] * 0.1                                   # GHz: cycles per 10 ns tick
print(f"lib {os.path.basename(_lib.LIB_PATH)}  shape B={B} {H}x{W} {Cin}->{Cout}  gap {gap_ms} ms  {launches} convs in {time.perf_counter() - t0:.1f} s, "
      f"{np.median(times):.4f} ms per conv (input transform + GEMM), {fl.value / (np.median(times) * 1e-3) / 1e12:.1f} TFLOP/s algorithmic")
print(f"  main loop of wino4_gemm_kernel, {len(a)} workgroups: {np.median(a[:, 0]):.0f} shader cycles = {np.median(a[:, 1]) * 10:.0f} ns (medians); "
      f"clock held: median {np.median(clk):.3f} GHz, p10 {np.quantile(clk, 0.1):.3f}, p90 {np.quantile(clk, 0.9):.3f}")
