#!/bin/bash
# GPU box, round 4 first call: whole -m gpu suite, default bench line, A/B of the 4-waves-per-SIMD build of the plain / pointwise
# direct conv instances (STCN_LIB=eva_vos_amd/csrc/build/exp/libstcn_hip_A.so) on the solo R1 kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4a
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
grep -E "^real" $O/bench_default.err
tail -c 6000 $O/bench_default.json
if [ -f $R/eva_vos_amd/csrc/build/exp/libstcn_hip_A.so ]; then
  bash tools/gpu_ab_trace.sh conv_gemm_kernel - "STCN_LIB=$R/eva_vos_amd/csrc/build/exp/libstcn_hip_A.so" 2>&1 | tee $O/ab_pw_waves.txt
fi
