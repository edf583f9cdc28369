#!/bin/bash
# GPU box: k-blocks per piece of the small F(4x4) launches (solo R1 kernel trace + one video in flight)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4n
rm -rf $O; mkdir -p $O
cd $R
bash tools/gpu_ab_trace.sh "wino4_gemm wino4_reduce" "STCN_WINO4_SMALL_KB=4" - "STCN_WINO4_SMALL_KB=16" 2>&1 | tee $O/ab_smallkb_trace.txt
