#!/bin/bash
# A/B by environment variable: bash tools/r06_ab_env.sh VAR val1 val2 -- "<workload args>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; shift 4
for wl in "$@"; do
  for rep in 1 2; do
    for v in $A $B; do
      OUT=$R/gpurun_out/abenv; rm -rf $OUT; mkdir -p $OUT
      env $VAR=$v rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing $wl > $OUT/bench.log 2>&1
      DB=$(find $OUT/kt -name "*.db" | head -1)
      python3 $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv > /dev/null
      echo "== $wl [$VAR=$v] rep $rep: $(grep '^{"metric"' $OUT/bench.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
      python3 -c "
import csv
for r in list(csv.reader(open('$OUT/kernel_stats.csv')))[1:8]:
    if 'rocclr' not in r[0] and 'at::' not in r[0]: print('   %-60s %6s %9s' % (r[0].replace('r4r::','')[:60], r[1], r[3]))"
      rm -rf $OUT/kt
    done
  done
done
