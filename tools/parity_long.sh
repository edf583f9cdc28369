#!/bin/bash
# GPU box: the two LONG parity legs that bench.py's default line only samples - a 24-round annotation session at 480p (default: 8 rounds)
# and BASELINE config 3 at its full length (T = 104, k = 5, every frame in the bank; default: 24 frames) - CPU oracle AND HIP engine, with the
# coded bounds of bench.py.  ~15 min of host time.  Writes gpurun_out/parity_long/{session24,config3_full}.json (copied to profiles/ by hand).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/parity_long
rm -rf $O; mkdir -p $O
cd $R
Q="--steps 4 --no-profile --no-r2 --no-memread-roofline --no-davis-val --no-drivers --no-session --value-repeats 1 --parity-long-frames 0"
python bench.py $Q --no-config3 --parity-session-rounds ${ROUNDS:-24} > $O/session.log 2> $O/session.err
python bench.py $Q --cpu-frames 0 --config3-oracle-frames ${C3FRAMES:-104} > $O/config3.log 2> $O/config3.err
python - <<PY
import json
def last(p): return json.loads([l for l in open(p) if l.startswith("{")][-1])
s = last("$O/session.log")["parity_session"]
json.dump(s, open("$O/session24.json", "w"), indent=1)
print(s["session"]); print("within_bound", s["within_bound"], "worst round", s["worst_round_mask_iou"], "worst frame", s["worst_round_min_frame_iou"], "same choice", s["same_frame_choice_every_round"])
for r in s["rounds"]:
    print(r["round"], r["frame"], round(r["mask_iou"], 6), r["mask_pixels_differing"], round(r["min_frame_iou"], 6), r["min_frame_iou_frame"], round(r["frame_bound"], 5), r["within_bound"], r["next_frame_oracle"], r["next_frame_hip"])
c = last("$O/config3.log")["config3"]["parity_vs_cpu_oracle"]
json.dump(c, open("$O/config3_full.json", "w"), indent=1)
print(c["sample"]); print("within_bound", c["within_bound"], "px differing", c["mask_pixels_differing"], "of", c["mask_pixels_total"], "decisive frac", c["decisive_pixel_fraction"], "ref self-noise", c.get("reference_self_noise"))
for o in c["per_object"]: print(o)
PY
# soak of the session logic: 3 x 48 seeded random sessions against the oracle (profiles/r05_soak.txt)
for seed in 7 8 9; do STCN_SOAK_SESSIONS=48 STCN_SOAK_SEED=$seed python -m pytest tests/test_gpu_sequence.py -m gpu -k random_annotation -q -n 4 --no-header -rf 2>&1 | tail -3; done
for seed in 3 4 5; do STCN_SOAK_MULTI=16 STCN_SOAK_SEED=$seed python -m pytest tests/test_gpu_sequence.py -m gpu -k random_multi -q -n 4 --no-header -rf 2>&1 | tail -3; done
for seed in 1 2; do STCN_SOAK_CONV=300 STCN_SOAK_MEMREAD=120 STCN_SOAK_SEED=$seed python -m pytest tests/test_gpu_kernels.py -m gpu -k "random_shapes or memory_read_matches" -q -n 4 --no-header -rf 2>&1 | tail -3; done
STCN_SOAK_SESSIONS=16 STCN_SOAK_480=1 STCN_SOAK_SEED=11 python -m pytest tests/test_gpu_sequence.py -m gpu -k random_annotation -q -n 4 --no-header -rf 2>&1 | tail -3
