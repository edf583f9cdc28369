#!/bin/bash
# GPU box: kernel trace of the decoder / key-encoder conv shapes (GEMM-only time per shape) + videos-in-flight sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o r -- python3 $R/tools/conv_shapes.py --batch 5 --iters 6 > $O/shapes.log 2>&1
cd $R
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python tools/trace_list.py $f > $O/trace_list.txt 2>&1
rm -rf $O/t
