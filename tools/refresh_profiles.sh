#!/bin/bash
# GPU box: collect the round's profile artifacts into gpurun_out/prof_final (then: python tools/make_profile_summaries.py r03)
#   1. default bench line (headline + roofline + config3 + memread roofline + cpu baseline), 2. one video in flight,
#   3. rocprofv3 kernel trace of the solo launches, 4. of the default command, 5. of the memory-read bench,
#   6./7. FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, no other trace domains).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
cd $R
SECONDS=0; python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench_default wall ${SECONDS} s" | tee $O/bench_default.wall
python bench.py --streams 1 --cpu-frames 0 --no-profile --no-r2 --no-config3 --no-memread-roofline --no-davis-val --no-drivers --no-session --no-power 2>/dev/null | tail -1 > $O/bench_streams1.json
cd /tmp && export TMPDIR=/tmp
Q="--cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline --no-davis-val --no-drivers --no-session --no-power --value-repeats 1"
STCN_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- python3 $R/bench.py --streams 1 --steps 2 --warmup 1 --no-profile $Q > $O/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o r -- python3 $R/bench.py $Q > $O/trace_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_memread -o r -- python3 $R/tools/memread_bench.py --k 5 > $O/trace_memread.log 2>&1
STCN_LOOKAHEAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o p -- python3 $R/bench.py --steps 1 --warmup 0 --streams 1 --no-profile --frames 30 $Q > $O/pmcF.log 2>&1
STCN_LOOKAHEAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o p -- python3 $R/bench.py --steps 1 --warmup 0 --streams 1 --no-profile --frames 30 $Q > $O/pmcW.log 2>&1
# rounds 2..8 of an annotation session (trace of 8 rounds minus trace of round 1), and the SQ / memory-side counters of the F(4x4) GEMM
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r2 -o r -- python3 $R/tools/r2_profile.py --no-class-profile > $O/trace_r2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_r1 -o r -- python3 $R/tools/r2_profile.py --no-class-profile --rounds 1 > $O/trace_r1.log 2>&1
python3 $R/tools/r2_profile.py > $O/r2_profile.txt 2>&1
cd $R && bash tools/pmc_f4.sh "256->256 @4" 5 > $O/pmc_wino4.txt 2>&1; cd /tmp
# round 4: matrix-pipe occupancy per kernel of the solo R1 leg (SQ counters, own pass), the memory read at k = 1 (what config 2 runs),
# and the F(4x4)-decoder A/B of pixels differing from the CPU oracle
STCN_LOOKAHEAD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcSQ -o p -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --no-profile $Q > $O/pmcSQ.log 2>&1
python3 $R/tools/mfma_busy.py $(find $O/pmcSQ -name "p_counter_collection.csv" | head -1) --md > $O/mfma_busy.md 2>&1
python3 $R/tools/memread_bench.py --k 1 > $O/memread_k1.txt 2>&1
python3 $R/tools/wino4_ab_parity.py > $O/wino4_ab_parity.txt 2>&1
find $O -name "r_kernel_trace.csv" -delete; find $O -name "p_kernel_trace.csv" -delete       # large, not needed for the summaries
python3 -c "import sys; sys.path.insert(0, '$R'); from eva_vos_amd import _lib; print(_lib.src_hash())" > $O/csrc_hash.txt 2>/dev/null
git -C $R rev-parse --short HEAD > $O/commit.txt 2>/dev/null || echo "${GRAFT_COMMIT:-unknown}" > $O/commit.txt
ls -la $O | head -30
