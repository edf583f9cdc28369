#!/bin/bash
# GPU box: A/B of environment knobs on the headline leg.  Usage: gpu_ab.sh "NAME=VAL ..." "NAME=VAL ..." (each arg = one arm; "-" = defaults)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
for rep in 1 2 3; do
  i=0
  for arm in "$@"; do
    i=$((i+1))
    for s in ${STREAMS:-4 1}; do
      if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
      env $envs python bench.py --streams $s --steps ${STEPS:-24} --warmup 4 --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline --no-davis-val --no-drivers --no-session --no-power --value-repeats 1 --no-profile > $O/b.json 2> $O/b.err
      python - <<PY
import json
l=[x for x in open("$O/b.json") if x.startswith("{")]
d=json.loads(l[-1]); print("rep $rep arm $i [$arm] streams $s: %.1f frames/s" % d["value"])
PY
    done
  done
done
