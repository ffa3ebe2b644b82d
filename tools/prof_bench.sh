#!/bin/bash
# kernel-trace + PMC passes over the default bench (the recipe behind profiles/): run on the MI355X box,
# bash tools/prof_bench.sh <tag> -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_summary.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
# BENCH_ARGS: e.g. "--workload cfg5_transnetpp_synthetic"; STEPS: "--steps 200 --warmup 20" profiles the default
# bench.py command (the first steps behind a fence run slower -- launch-queue fill, clocks: tools/steps_probe.py --
# so a 20-step trace averages ~5 % above the steady state the default line reports)
STEPS=${STEPS:-"--steps 20 --warmup 5"}
CMD="python $R/bench.py $STEPS --no-cpu-baseline $BENCH_ARGS"
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $CMD > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.log 2>&1
python3 $R/tools/pmc_summary.py $OUT $OUT/pmc_summary.json "bench.py $STEPS $BENCH_ARGS, native engine"
cat $OUT/kernel_stats.csv
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/kt
