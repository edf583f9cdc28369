#!/bin/bash
# GPU box: SQ counters of the memory-read kernels on one bank shape.  usage: bash tools/pmc_memread.sh T,Q
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_memread; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s$i -o p -- python3 $R/tools/memread_bench.py --only "$1" > $O/s$i.log 2>&1
  for k in "affinity_tile_kernel<true>" "affinity_tile_kernel<false>"; do echo "== $k"; python3 $R/tools/pmc_summary.py $O/s$i/p_counter_collection.csv --kernel "$k"; done
done
