#!/bin/bash
# GPU box, round 4: whole -m gpu suite + smoke, then the round's profile artifacts (tools/refresh_profiles.sh -> gpurun_out/prof_final)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4b
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf --durations=12 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/refresh_profiles.sh > $O/refresh.log 2>&1
tail -5 $O/refresh.log
tail -c 1500 gpurun_out/prof_final/bench_default.json
cat gpurun_out/prof_final/wino4_ab_parity.txt gpurun_out/prof_final/memread_k1.txt
