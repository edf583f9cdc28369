#!/bin/bash
# GPU box: SQ counters of the conv kernels on one conv shape.  usage: bash tools/pmc_wino.sh "<conv_shapes --only pattern>" [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_wino; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s$i -o p -- python3 $R/tools/conv_shapes.py --only "$1" --batch ${2:-5} --iters 3 > $O/s$i.log 2>&1
  for k in "wino_gemm_kernel" "conv_gemm_kernel"; do echo "== $k"; python3 $R/tools/pmc_summary.py $O/s$i/p_counter_collection.csv --kernel "$k"; done
done
grep -E "^(key|dec|val|custom)" $O/s1.log
