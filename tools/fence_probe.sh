#!/bin/bash
# What the first steps behind bench.py's fence look like on the device: kernel trace of the driver's 20-step command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fence_$1
mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/kt -- python $R/bench.py ${STEPS:---steps 20 --warmup 5} --no-cpu-baseline $BENCH_ARGS > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv > /dev/null; head -7 $OUT/kernel_stats.csv
python3 $R/tools/kernel_timeline.py $DB ${ROWS:-12} 40 > $OUT/timeline.txt
cat $OUT/timeline.txt
rm -rf $OUT/kt
