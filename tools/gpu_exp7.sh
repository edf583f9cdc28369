#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -x -k "conv" 2>&1 | tail -2
for i in 1 2 3; do python tools/conv_shapes.py --batch 1 --only "fuse 3x3" 2>&1 | grep "fuse"; done
python tools/r2_profile.py 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | tail -11
