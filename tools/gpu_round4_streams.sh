#!/bin/bash
# GPU box: videos in flight per GPU on the headline leg (final build of round 4)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
STREAMS="${STREAMS_SWEEP:-4 6 8 10 12}" STEPS=${STEPS:-48} bash tools/gpu_ab.sh - 2>&1 | tee $O/ab_streams_${STEPS:-48}.txt
