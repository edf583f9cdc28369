#!/bin/bash
# GPU box: the pointwise chain kernel - its tests, then A/B on the solo R1 kernel trace and on the headline leg
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -x -k "chain or conv_matches or random_shapes" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
bash tools/gpu_ab_trace.sh "conv_gemm_kernel pw_chain_kernel" - "STCN_PW_CHAIN=1" 2>&1 | tee $O/ab_chain_trace.txt
STREAMS="4" STEPS=24 bash tools/gpu_ab.sh - "STCN_PW_CHAIN=1" 2>&1 | tee $O/ab_chain_bench.txt
