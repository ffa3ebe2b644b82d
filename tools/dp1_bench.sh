#!/bin/bash
# The data-parallel step through RCCL at ONE rank (R4R_DP_SINGLE=1: every collective is issued, nothing crosses a
# wire): what the exchange MACHINERY costs next to the single-process step.  bash tools/dp1_bench.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']
print('%-28s %9.0f ratings/s %8.4f ms/step (gpu %.4f)  exchange %s %s' % ('$1', d['value'], d['ms_per_step'], d.get('gpu_ms_per_step', 0), c.get('dp_exchange'), c.get('dp_exchange_ms')))"; }
for rep in 1 2; do
  python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | line "single process"
  for rccl in 1 0; do for ex in allreduce gather; do
    R4R_DP_SINGLE=1 R4R_DP_RCCL=$rccl R4R_DP_EXCHANGE=$ex python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      $R/bench.py --gpus 1 --no-cpu-baseline --strong-leg "" "$@" 2>/dev/null | line "dp1 stream_rccl=$rccl $ex"
  done; done
  R4R_DP_SINGLE=1 R4R_DP_EXCHANGE=peer python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      $R/bench.py --gpus 1 --no-cpu-baseline --strong-leg "" "$@" 2>/dev/null | line "dp1 peer-mapped exchange"
done
