#!/bin/bash
# GPU box: engine knobs on the headline leg (4 videos in flight) after the chain kernel: side-stream look-ahead, key-encoder batch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4f
rm -rf $O; mkdir -p $O
cd $R
STREAMS="4" STEPS=24 bash tools/gpu_ab.sh - "STCN_LOOKAHEAD=2" "STCN_KEY_BATCH=8" 2>&1 | tee $O/ab_knobs.txt
python -m pytest tests/test_gpu_sequence.py -m gpu -q --no-header -k "engine_options or seq480k5" -s 2>&1 | grep -E "HIP vs golden seq480k5|passed|failed" | tee $O/pytest.txt
