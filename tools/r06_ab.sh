#!/bin/bash
# Same-box A/B of two builds: kernel-trace averages per workload.  bash tools/r06_ab.sh <tagA>:<libA> <tagB>:<libB> -- <workload args>...
# e.g. bash tools/r06_ab.sh old:libr4r_hip_var_old.so new:libr4r_hip.so -- "--workload cfg4_narre_kindle" "--workload cfg2_mfdot_electronics"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
libs=()
while [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
for wl in "$@"; do
  for rep in 1 2; do
    for tl in "${libs[@]}"; do
      tag=${tl%%:*}; lib=${tl##*:}
      OUT=$R/gpurun_out/ab_$tag
      rm -rf $OUT; mkdir -p $OUT
      R4R_LIBRARY=$R/reviews4rec_amd/csrc/$lib rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing $wl > $OUT/bench.log 2>&1
      DB=$(find $OUT/kt -name "*.db" | head -1)
      python3 $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv > /dev/null
      echo "== $wl [$tag] rep $rep: $(grep '^{"metric"' $OUT/bench.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
      python3 -c "
import csv,sys
for r in list(csv.reader(open('$OUT/kernel_stats.csv')))[1:8]:
    print('   %-60s %6s %9s' % (r[0].replace('r4r::','')[:60], r[1], r[3]))"
      rm -rf $OUT/kt
    done
  done
done
