#!/bin/bash
# GPU box: SQ-side counters (instruction mix, MFMA-busy, LDS conflicts, clock) of the kernels of one conv shape.
# usage: bash tools/pmc_sq.sh "<conv_shapes --only pattern>" "<kernel name substring>" [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s$i -o p -- python3 $R/tools/conv_shapes.py --only "$1" --batch ${3:-5} --iters 3 > $O/s$i.log 2>&1
  python3 $R/tools/pmc_summary.py $O/s$i/p_counter_collection.csv --kernel "$2"
done
grep -E "^(key|dec|val|custom|fuse)" $O/s1.log
find $O -name "p_kernel_trace.csv" -delete
