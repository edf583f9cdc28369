#!/bin/bash
# GPU box: collect the round's profile artifacts into gpurun_out/prof_final (then: python tools/make_profile_summaries.py)
#   1. default bench line, 2. config-3 bench line, 3. rocprofv3 kernel trace of the solo launches,
#   4./5. FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, no other trace domains).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --objects 3 --mem-freq 1 --frames 40 --steps 3 --warmup 1 --streams 1 --cpu-frames 0 --no-f16x3-leg 2>/dev/null | tail -1 > $O/bench_config3.json
python bench.py --streams 1 --cpu-frames 0 --no-f16x3-leg --no-profile --no-r2 2>/dev/null | tail -1 > $O/bench_streams1.json
cd /tmp && export TMPDIR=/tmp
STCN_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- python3 $R/bench.py --streams 1 --steps 2 --warmup 1 --cpu-frames 0 --no-profile --no-f16x3-leg --no-r2 > $O/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o r -- python3 $R/bench.py --cpu-frames 0 --no-f16x3-leg --no-r2 > $O/trace_default.log 2>&1
STCN_LOOKAHEAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o p -- python3 $R/bench.py --steps 1 --warmup 0 --streams 1 --cpu-frames 0 --no-profile --no-f16x3-leg --no-r2 --frames 30 > $O/pmcF.log 2>&1
STCN_LOOKAHEAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o p -- python3 $R/bench.py --steps 1 --warmup 0 --streams 1 --cpu-frames 0 --no-profile --no-f16x3-leg --no-r2 --frames 30 > $O/pmcW.log 2>&1
rm -f $O/*/r_kernel_trace.csv $O/*/p_kernel_trace.csv          # large, not needed for the summaries
ls -la $O $O/trace $O/pmcF | head -30
