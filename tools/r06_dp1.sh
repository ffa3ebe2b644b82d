#!/bin/bash
# One-rank RCCL job (R4R_DP_SINGLE=1) of the ID-table configurations next to the single-process step: what the
# data-parallel MACHINERY costs (profiles/r06_rccl1.log).  bash tools/r06_dp1.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-72s %9.0f ratings/s %8.4f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
dp1() { R4R_DP_SINGLE=1 "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      $R/bench.py --gpus 1 --no-cpu-baseline --no-config-legs --strong-leg "" $ARGS 2>/dev/null; }
for rep in 1 2; do
  ARGS="--workload cfg2_mfdot_electronics"
  python $R/bench.py --no-cpu-baseline --no-config-legs $ARGS 2>/dev/null | line "cfg2 single process (2 launches)"
  dp1 env | line "cfg2 dp1: grad, RCCL all_gather, update scanning the ids (3 launches)"
  dp1 env R4R_MF_DP_REGISTER=1 | line "cfg2 dp1: ... with the registering launch of round 4 (4 launches)"
  dp1 env R4R_DP_EXCHANGE=peer | line "cfg2 dp1: R4R_DP_EXCHANGE=peer, no collective (2 launches)"
  ARGS="--workload cfg5_transnetpp_synthetic"
  python $R/bench.py --no-cpu-baseline --no-config-legs $ARGS 2>/dev/null | line "cfg5 single process (6 launches)"
  dp1 env | line "cfg5 dp1: one block, one all_gather, update over the blocks"
  ARGS="--workload cfg4_narre_kindle"
  python $R/bench.py --no-cpu-baseline --no-config-legs $ARGS 2>/dev/null | line "cfg4 single process (5 launches)"
  dp1 env | line "cfg4 dp1: one block, one all_gather, entry waves + sweep over the blocks"
  ARGS=""
  python $R/bench.py --no-cpu-baseline --no-config-legs $ARGS 2>/dev/null | line "cfg3 single process (5 launches)"
  dp1 env | line "cfg3 dp1 (exchange autotuned)"
  for spec in "cfg2_mfdot_electronics MF" "cfg2_mfdot_electronics NeuMF" "cfg3_deepconn_electronics_e300 deepconn++"; do
    set -- $spec
    ARGS="--workload $1 --model-type $2"
    python $R/bench.py --no-cpu-baseline --no-config-legs $ARGS 2>/dev/null | line "$2 (shapes of $1) single process"
    dp1 env | line "$2 dp1: pack, one all_gather, unpack, update"
  done
done
