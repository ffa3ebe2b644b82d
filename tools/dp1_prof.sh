#!/bin/bash
# kernel stats of the one-rank data-parallel step under one exchange form: bash tools/dp1_prof.sh peer|allreduce|gather
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/dp1prof_$1
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 R4R_DP_SINGLE=1 R4R_DP_EXCHANGE=$1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 300)) RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $R/bench.py --gpus 1 --no-cpu-baseline --strong-leg "" --no-config-legs $DP1_ARGS > $OUT/bench.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
python3 $R/tools/kernel_timeline.py $DB 14 40 | tail -18
rm -rf $OUT/kt
