#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |pytest rc" $O/pytest.log | head -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "--- dual sweep A/B"
python tools/dual_sweep_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/dual_sweep_ab.txt
echo "--- driver lanes + host account"
python tools/driver_lanes.py 8 40 2>&1 | grep -v amdgpu.ids | tee $O/driver_lanes.txt
echo "--- bench (sessions + drivers only)"
python bench.py --steps 8 --cpu-frames 0 --no-config3 --no-memread-roofline --no-davis-val --no-power --value-repeats 1 > $O/bench_short.json 2> $O/bench_short.err
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_short.json") if l.startswith("{")][-1])
print("value", d["value"], "r2 one video", d["roofline_r2"]["frames_per_s_one_video"], "solo", d["roofline_r2"]["frames_per_s_solo"], "in flight", d["roofline_r2"]["frames_per_s_videos_in_flight"])
for k, v in d["session"].items():
    print(k, {l: (round(x["rounds_per_s"], 1), round(x["propagated_frames_per_s"]), round(x["device_busy_frac"], 3)) for l, x in v["lanes"].items()}, "frames/round", round(v["frames_per_round_mean"], 2), "kernel ms/frame", round(v["kernel_ms_per_propagated_frame"], 3))
print(json.dumps(d["drivers"]))
PY
