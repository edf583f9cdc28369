#!/bin/bash
# GPU box, round 6 batch b: new driver path + second key stream + knob A/Bs + the in-loop clock experiment
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6b
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --no-header -rf -k "round_scorer or fq_driver or eval_driver or driver or two_key_encoder or key_batching or pinned or selectors or oracle_mask or conv_matches or slow_side or bench" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |pytest rc" $O/pytest.log | head -30
echo "--- A/B headline: key streams 1 vs 2 (one video in flight), key_proj on F(4x4)"
STREAMS=1 STEPS=12 bash tools/gpu_ab.sh "STCN_KEY_STREAMS=1" "STCN_KEY_STREAMS=2" "STCN_WINO4_KEYPROJ=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_keystreams.txt
echo "--- A/B headline 4 lanes: key_proj on F(4x4)"
STREAMS=4 STEPS=24 bash tools/gpu_ab.sh "-" "STCN_WINO4_KEYPROJ=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_keyproj.txt
echo "--- key_proj F(4x4) parity A/B"
AB_ENV="STCN_WINO4_KEYPROJ=1" python tools/wino4_ab_parity.py 2>&1 | grep -v amdgpu.ids | tee $O/keyproj_parity.txt
echo "--- session breakdown 60 rounds"
python tools/session_breakdown.py --frames 66 --rounds 60 --metric j_and_f 2>&1 | grep -v amdgpu.ids > $O/breakdown60.txt; grep -E "^---|kernel ms total" $O/breakdown60.txt
echo "--- driver lanes"
python tools/driver_lanes.py 8 40 2>&1 | grep -v amdgpu.ids | tee $O/driver_lanes.txt
echo "--- clock"
bash tools/w4_clock.sh
cp $R/gpurun_out/w4_clock/clock.txt $O/ 2>/dev/null
