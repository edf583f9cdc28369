#!/bin/bash
# GPU box: A/B of environment knobs on the headline AND the in-flight second-interaction rate.  usage: gpu_ab_r2.sh "NAME=VAL" ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
  for arm in "$@"; do
    if [ "$arm" = "-" ]; then envs="X_=1"; else envs="$arm"; fi
    env $envs python bench.py --steps 24 --warmup 4 --cpu-frames 0 --no-config3 --no-memread-roofline --no-davis-val --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rep $rep [$arm]: R1 %.1f  R2 %.1f frames/s' % (d['value'], d['r2_frames_per_s_rank0']))"
  done
done
