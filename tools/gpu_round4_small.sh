#!/bin/bash
# GPU box: F(4x4) for small launches (every tile in K pieces): conv tests, the value-encoder shapes, the whole suite, A/B against the previous build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4m
rm -rf $O; mkdir -p $O
cd $R
python tools/conv_shapes.py --only val 2>&1 | grep -E "^val|weighted" | tee $O/val_shapes.txt
timeout 2400 python -m pytest tests -m gpu -q --no-header > $O/pytest.log 2>&1
tail -5 $O/pytest.log
bash tools/gpu_ab_trace.sh "wino_gemm wino4_gemm wino_input wino4_input reduce" "STCN_LIB=$R/eva_vos_amd/csrc/build/exp/libstcn_hip_prev.so" - 2>&1 | tee $O/ab_small_trace.txt
STREAMS="4 1" STEPS=48 bash tools/gpu_ab.sh "STCN_LIB=$R/eva_vos_amd/csrc/build/exp/libstcn_hip_prev.so" - 2>&1 | tee $O/ab_small_bench.txt
