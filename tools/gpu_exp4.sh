#!/bin/bash
# GPU box: F(4x4) GEMM time per round of workgroups against the size of V (cache residency): 256 -> 256 at 256x128, B = 1, 2, 4, 8
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp4
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1 2 4 8; do
  STCN_BENCH_CONV_F4=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -o r -- python3 $R/tools/conv_shapes.py --shape $B,256,128,256,256,3,1 --iters 6 > $O/b$B.log 2>&1
  grep custom $O/b$B.log
  python $R/tools/kstat.py $O/b$B wino4
  find $O/b$B -name "*kernel_trace.csv" -delete
done
