#!/bin/bash
# GPU box: with several videos in flight, is split-K of the small F(2x2) GEMMs (which exists to fill the chip for ONE video) still worth its
# reduce launches?  Headline leg (4 streams) under STCN_WINO_SPLIT_BELOW = 160 (default) / 64 / 0, then one stream.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e
rm -rf $O; mkdir -p $O
cd $R
STREAMS="4 1" STEPS=24 bash tools/gpu_ab.sh - "STCN_WINO_SPLIT_BELOW=64" "STCN_WINO_SPLIT_BELOW=0" 2>&1 | tee $O/ab_splitk.txt
