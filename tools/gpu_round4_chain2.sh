#!/bin/bash
# GPU box: the two tile walks of the pointwise chain kernel (1 consecutive, 2 strided) against the one-tile instance: tests, kernel trace, headline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4g
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -k "chain" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
bash tools/gpu_ab_trace.sh "conv_gemm_kernel pw_chain_kernel" "STCN_PW_CHAIN=0" "STCN_PW_CHAIN=1" "STCN_PW_CHAIN=2" 2>&1 | tee $O/ab_chain_trace.txt
STREAMS="4" STEPS=24 bash tools/gpu_ab.sh "STCN_PW_CHAIN=0" "STCN_PW_CHAIN=1" "STCN_PW_CHAIN=2" "STCN_PW_CHAIN=2 STCN_KEY_BATCH=8" 2>&1 | tee $O/ab_chain_bench.txt
