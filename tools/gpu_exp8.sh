#!/bin/bash
# GPU box: per-(kernel, grid) launch table of the solo R1 leg
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp8
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STCN_LOOKAHEAD=0 rocprofv3 --kernel-trace --output-format csv -d $O/t -o r -- python3 $R/bench.py --streams 1 --steps 1 --warmup 1 --no-profile --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline --no-davis-val > $O/log 2>&1
cd $R
python tools/trace_list.py $(find $O/t -name "*kernel_trace.csv") > $O/trace_list.txt
rm -rf $O/t
