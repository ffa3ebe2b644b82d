#!/bin/bash
# One-rank RCCL job (R4R_DP_SINGLE=1) of the ID-table configurations next to the single-process step: what the
# data-parallel MACHINERY costs (profiles/r05_rccl1.log).  bash tools/r05_dp1.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-46s %9.0f ratings/s %8.4f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
dp1() { R4R_DP_SINGLE=1 "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      $R/bench.py --gpus 1 --no-cpu-baseline --no-config-legs --strong-leg "" $ARGS 2>/dev/null; }
for wl in cfg2_mfdot_electronics cfg5_transnetpp_synthetic; do
  ARGS="--workload $wl"
  for rep in 1 2; do
    python $R/bench.py --no-cpu-baseline --no-config-legs $ARGS 2>/dev/null | line "$wl single process"
    dp1 env | line "$wl dp1 (rows found by scanning the ids)"
    dp1 env R4R_MF_DP_REGISTER=1 | line "$wl dp1, registered form (round 4)"
    dp1 env R4R_DP_EXCHANGE=peer | line "$wl dp1, peer-mapped exchange (no collective)"
  done
done
