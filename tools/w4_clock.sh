#!/bin/bash
# GPU box: the F(4x4) GEMM's in-loop clock - watts or feed?  Three rows -> gpurun_out/w4_clock/clock.txt
#   (i) shipped loop, back to back for 2.5 s   (ii) shipped loop, 3 ms idle gap after every launch   (iii) MFMAs replaced by 16 v_fma each
# Needs the diagnostic builds (made in the build container, they travel with the snapshot):
#   make -C eva_vos_amd/csrc EXTRA=-DSTCN_W4_CLOCK=1 OBJDIR=build/objclk1 OUT=build/exp/libstcn_hip_clk1.so   (and =2 -> clk2)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/w4_clock
mkdir -p $O
cd $R
E=$R/eva_vos_amd/csrc/build/exp
{
for shape in 5,120,216,256,256 5,60,108,512,256; do
  echo "=== shape $shape"
  echo "(i) shipped loop, back to back";        STCN_LIB=$E/libstcn_hip_clk1.so python tools/w4_clock.py --shape $shape
  echo "(ii) shipped loop, 3 ms idle gaps";      STCN_LIB=$E/libstcn_hip_clk1.so python tools/w4_clock.py --shape $shape --gap-ms 3
  echo "(iii) v_fma instead of MFMA, back to back"; STCN_LIB=$E/libstcn_hip_clk2.so python tools/w4_clock.py --shape $shape
  echo "(iii') v_fma instead of MFMA, 3 ms gaps";   STCN_LIB=$E/libstcn_hip_clk2.so python tools/w4_clock.py --shape $shape --gap-ms 3
done
echo "=== register-operand MFMA probe (no memory traffic) for reference"
python - <<PY
import ctypes as C, torch, sys
sys.path.insert(0, "$R")
from eva_vos_amd import _lib
tf, ms = C.c_float(), C.c_float()
_lib.check(_lib.lib().stcn_bench_mfma_rate(C.c_void_p(torch.cuda.current_stream().cuda_stream), 500, C.byref(tf), C.byref(ms)))
print(f"mfma_probe_kernel 500 ms: {tf.value:.1f} TFLOP/s = {tf.value / 157.3 * 2.4:.3f} GHz-equivalent")
PY
} 2>&1 | grep -v "amdgpu.ids" | tee $O/clock.txt
