#!/bin/bash
# GPU box: -m gpu suite + rounds-2..8 profile + per-shape conv table (args: extra pytest -k expression)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT:-quick}
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf --durations=8 ${1:+-k "$1"} > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.log | head -40
python tools/r2_profile.py > $O/r2_profile.txt 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/r2_profile.txt | tail -14
python tools/conv_shapes.py --batch 5 > $O/conv_shapes_b5.txt 2>&1
tail -4 $O/conv_shapes_b5.txt
