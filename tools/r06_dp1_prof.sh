#!/bin/bash
# one-rank RCCL job (R4R_DP_SINGLE=1) under the kernel trace: which launches the data-parallel step consists of and how long they take
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for wl in "--workload cfg2_mfdot_electronics" "--workload cfg4_narre_kindle" "--workload cfg2_mfdot_electronics --model-type MF" ""; do
  OUT=$R/gpurun_out/dp1prof; rm -rf $OUT; mkdir -p $OUT
  R4R_DP_SINGLE=1 rocprofv3 --kernel-trace --stats -d $OUT/kt -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) $R/bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-config-legs --no-kernel-timing --strong-leg "" $wl > $OUT/bench.log 2>&1
  echo "== dp1 $wl: $(grep '^{"metric"' $OUT/bench.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  for DB in $(find $OUT/kt -name "*.db"); do
    python3 $R/tools/rocpd_stats.py $DB $OUT/k.csv > /dev/null
    python3 -c "
import csv
for r in list(csv.reader(open('$OUT/k.csv')))[1:12]:
    if int(r[1]) >= 400: print('   %-70s %6s %9s' % (r[0].replace('r4r::','')[:70], r[1], r[3]))"
  done
done
