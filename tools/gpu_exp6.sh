#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp6
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -x -k "conv" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for fw in 1 0; do
  echo "== STCN_FUSION_WINO=$fw"
  STCN_FUSION_WINO=$fw python tools/conv_shapes.py --batch 1 --only "fuse 3x3 32" 2>&1 | grep "fuse"
done
python tools/r2_profile.py 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | tail -12
