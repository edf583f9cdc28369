#!/bin/bash
# GPU box: masks returned through pinned host memory vs the plain .cpu(): cost per interact, one video in flight, headline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4o
rm -rf $O; mkdir -p $O
cd $R
for v in 0 1; do echo "STCN_PINNED_DOWNLOAD=$v"; STCN_PINNED_DOWNLOAD=$v python tools/download_cost.py 66 2>&1 | grep -E "interact|cpu"; done | tee $O/download_cost.txt
STREAMS="4 1" STEPS=24 bash tools/gpu_ab.sh "STCN_PINNED_DOWNLOAD=0" - 2>&1 | tee $O/ab_pinned.txt
python -m pytest tests/test_gpu_sequence.py -m gpu -q --no-header -k "goldens or deepcopy or stream or inputs_on" 2>&1 | tail -2
