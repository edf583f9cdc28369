#!/bin/bash
# GPU box, end of round 4: the -m gpu suite with NaN-poisoned recycled buffers, then what the driver runs (suite, smoke, default bench line)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4k
rm -rf $O; mkdir -p $O
cd $R
STCN_POOL_POISON=1 timeout 2400 python -m pytest tests -m gpu -q --no-header -x -k "not bench and not session" > $O/pytest_poison.log 2>&1
tail -3 $O/pytest_poison.log
bash tools/final_run.sh 2>&1 | tail -12
