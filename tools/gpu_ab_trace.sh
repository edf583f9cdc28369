#!/bin/bash
# GPU box: A/B of environment knobs on the kernel-time sum of the solo R1 leg (rocprofv3 --stats), arms alternating.
# usage: gpu_ab_trace.sh "<kernel substring>" "NAME=VAL ..." "NAME=VAL ..."   ("-" = defaults)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/abt
rm -rf $O; mkdir -p $O
K="$1"; shift
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  i=0
  for arm in "$@"; do
    i=$((i+1))
    if [ "$arm" = "-" ]; then envs="X_=1"; else envs="$arm"; fi
    export $envs
    STCN_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/a${i}_$rep -o r -- python3 $R/bench.py --streams 1 --steps 2 --warmup 1 --no-profile --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline --no-davis-val --no-drivers --no-session --no-power --value-repeats 1 > $O/log 2>&1
    for e in $envs; do unset ${e%%=*}; done
    echo "rep $rep arm $i [$arm]: $(python3 $R/tools/kstat.py $O/a${i}_$rep $K | awk -v k="$K" 'NR==1{t=$4} NR>1{s+=$1} END{printf "total %s ms, %s kernels %.2f ms", t, k, s}')"
    find $O/a${i}_$rep -name "*kernel_trace.csv" -delete
  done
done
