#!/bin/bash
# GPU box: memory-side counters of the Winograd GEMMs on one conv shape.  usage: bash tools/pmc_f4.sh "<conv_shapes --only pattern>" [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_f4; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  STCN_BENCH_CONV_F4=${F4:-1} timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/s$i -o p -- python3 $R/tools/conv_shapes.py --only "$1" --batch ${2:-5} --iters 3 > $O/s$i.log 2>&1
  for k in "wino4_gemm_kernel" "wino_gemm_kernel" "wino4_input" ; do echo "== $k"; python3 $R/tools/pmc_summary.py $O/s$i/p_counter_collection.csv --kernel "$k"; done
done
find $O -name "p_kernel_trace.csv" -delete
