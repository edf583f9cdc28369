#!/bin/bash
# GPU box: whole -m gpu suite + default bench line (state check after a change)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4d
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
grep -E "^real" $O/bench_default.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]; c=d["config3"]; p=d["parity_vs_cpu_oracle"]
print("value", d["value"], "r2", d["r2_frames_per_s_rank0"], "frac", r["frac"], "frame_frac", r["frame_executed_frac"], "frame_ms", r["frame_kernel_ms"])
print("config3", c["frames_per_s"], c["mask_iou_vs_cpu_oracle"], "parity r1", p["mask_iou_hip_vs_cpu_oracle_r1"], p["min_frame_iou_hip_vs_cpu_oracle_r1"], "r2", p["mask_iou_hip_vs_cpu_oracle_r2"], p["min_frame_iou_hip_vs_cpu_oracle_r2"])
print("long", d["parity_long_clip"]["mask_iou_hip_vs_cpu_oracle"], d["parity_long_clip"]["min_frame_iou"], "session", d["parity_session"]["worst_round_mask_iou"], d["parity_session"]["worst_round_min_frame_iou"])
print("identical", d["concurrent_videos_bit_identical"], "davis", d["davis_val"]["frames_per_s"])
PY
