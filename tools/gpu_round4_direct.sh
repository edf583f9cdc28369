#!/bin/bash
# GPU box: unfused sweeps decoding straight into prob + the fused bank key copy: whole gpu suite, then A/B against the previous build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4i
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --no-header -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
bash tools/gpu_ab_trace.sh "copy copyBuffer" "STCN_LIB=$R/eva_vos_amd/csrc/build/exp/libstcn_hip_prev.so" - 2>&1 | tee $O/ab_direct_trace.txt
STREAMS="4" STEPS=24 bash tools/gpu_ab.sh "STCN_LIB=$R/eva_vos_amd/csrc/build/exp/libstcn_hip_prev.so" - 2>&1 | tee $O/ab_direct_bench.txt
