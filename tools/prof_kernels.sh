#!/bin/bash
# kernel-trace pass only (per-kernel time): bash tools/prof_kernels.sh <tag>   (BENCH_ARGS as in prof_bench.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing $BENCH_ARGS"
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $CMD > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv
rm -rf $OUT/kt
