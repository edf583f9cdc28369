#!/bin/bash
# GPU box: (1) key-encoder layer times at a batch of 5 and of 10 frames (is a key pass over two decode groups worth building?),
# (2) BASELINE config 3 at full length against the CPU oracle
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4j
rm -rf $O; mkdir -p $O
cd $R
python tools/conv_shapes.py --only key --batch 5 > $O/key_b5.txt 2>&1
python tools/conv_shapes.py --only key --batch 10 > $O/key_b10.txt 2>&1
paste <(awk '{print $1,$2,$3,$4,$(NF-2)}' $O/key_b5.txt) <(awk '{print $(NF-2)}' $O/key_b10.txt) | column -t
bash tools/gpu_round4_cfg3_full.sh
