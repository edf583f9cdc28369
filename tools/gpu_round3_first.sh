#!/bin/bash
# GPU box, round 3 first call: whole -m gpu suite, default bench line, R2 kernel trace, per-shape conv table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --no-header -rf --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
python tools/conv_shapes.py --batch 5 > $O/conv_shapes_b5.txt 2>&1
python tools/conv_shapes.py > $O/conv_shapes_b1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/r2_profile.py > $O/r2_profile.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r2 -o r -- python3 $R/tools/r2_profile.py --no-class-profile > $O/trace_r2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r1 -o r -- python3 $R/tools/r2_profile.py --no-class-profile --rounds 1 > $O/trace_r1.log 2>&1
find $O -name "r_kernel_trace.csv" -delete
cat $O/r2_profile.txt | tail -30
