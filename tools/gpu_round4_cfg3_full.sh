#!/bin/bash
# GPU box: BASELINE config 3 at FULL length on the CPU oracle AND the HIP engine (k = 5, mem_freq = 1, T = 104: the bank grows to 168 480 rows,
# the memory read runs its sampled plans ss = 1 .. 8 end to end) - about 6 minutes of host time for the oracle, once per round
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4h
rm -rf $O; mkdir -p $O
cd $R
( time python bench.py --config3-oracle-frames 104 --no-davis-val --no-r2 --cpu-frames 0 --no-memread-roofline --no-profile --steps 4 ) > $O/bench_cfg3_full.json 2> $O/bench_cfg3_full.err
grep -E "^real" $O/bench_cfg3_full.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_cfg3_full.json") if l.startswith("{")][-1])
c=d["config3"]; p=c["parity_vs_cpu_oracle"]
print(json.dumps({k:p[k] for k in p if k!="what"}, indent=1))
print("fps", c["frames_per_s"])
PY
