#!/bin/bash
# GPU box: conv kernel tests + per-shape table (with kernel trace) + default-shaped bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT:-exp2}
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -x -k "conv or wino" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o r -- python3 $R/tools/conv_shapes.py --batch 5 --iters 6 > $O/shapes.log 2>&1
cd $R
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python tools/trace_list.py $f > $O/trace_list.txt 2>&1
rm -rf $O/t
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/shapes.log | grep "dec\|key_comp\|fuser\|weighted"
grep "wino4_gemm" $O/trace_list.txt
python bench.py --steps 24 --warmup 4 --cpu-frames 0 --no-config3 --no-memread-roofline --no-davis-val > $O/bench.json 2> $O/bench.err
python - <<PY
import json
l=[x for x in open("$O/bench.json") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "r2", d.get("r2_frames_per_s_rank0"), "roof", d["roofline"]["frac"] if d.get("roofline") else None)
PY
