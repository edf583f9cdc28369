#!/bin/bash
# GPU box: kernel tests, then the rocprofv3 kernel-time sum of the solo R1 leg (4 videos x 65 frames) and of rounds 2..8
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT:-exp3}
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
STCN_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r1 -o r -- python3 $R/bench.py --streams 1 --steps 2 --warmup 1 --no-profile --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline --no-davis-val > $O/r1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2 -o r -- python3 $R/tools/r2_profile.py --no-class-profile > $O/r2.log 2>&1
cd $R
python tools/kstat.py $O/r1 gemm input reduce
python tools/kstat.py $O/r2 gemm input fusion
find $O -name "*kernel_trace.csv" -delete
grep "rounds 2" $O/r2.log
