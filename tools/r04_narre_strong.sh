#!/bin/bash
# bench.py --workload cfg4_narre_kindle --scaling strong --global-batch 8192 on the native step at
# one rank and at two ranks sharing one GPU (gloo): bash tools/r04_narre_strong.sh > gpurun_out/r04_narre_strong.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--workload cfg4_narre_kindle --scaling strong --global-batch ${GB:-8192} --steps 20 --warmup 5 --ramp 10 --no-cpu-baseline --pool 2"
line() { python3 -c "
import sys, json
t = sys.stdin.read().strip().splitlines()
d = json.loads(t[-1])
c = d['config']
print('%-10s %10.0f ratings/s %9.4f ms/step  engine %s  batch/gpu %d  replicas_identical %s' % ('$1', d['value'], d['ms_per_step'], c['engine'], c['batch_per_gpu'], c.get('replicas_identical')))"; }
python $R/bench.py $ARGS 2>$R/gpurun_out/narre_strong_1.err | line "1 rank"
R4R_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
  $R/bench.py --gpus 2 $ARGS 2>$R/gpurun_out/narre_strong_2.err | line "2 ranks"
tail -3 $R/gpurun_out/narre_strong_1.err $R/gpurun_out/narre_strong_2.err
