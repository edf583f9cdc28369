#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for mc in 128 64; do
  echo "== STCN_WINO_MIN_CIN=$mc"
  STCN_WINO_MIN_CIN=$mc python tools/conv_shapes.py --batch 5 --only "3x3 64" 2>&1 | grep "3x3 64"
  STCN_WINO_MIN_CIN=$mc python tools/conv_shapes.py --batch 1 --only "val.l1" 2>&1 | grep "3x3 64"
done
