#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of one conv shape.  usage: bash tools/pmc_conv.sh "<conv_shapes --only pattern>" [extra conv_shapes args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O
PAT="$1"; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o p -- python3 $R/tools/conv_shapes.py --only "$PAT" --iters 3 "$@" > $O/$c.log 2>&1
  python3 $R/tools/pmc_summary.py $O/$c/p_counter_collection.csv --kernel conv_gemm
done
grep -E "^(key|dec|val|custom)" $O/FETCH_SIZE.log
