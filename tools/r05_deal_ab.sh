#!/bin/bash
# gather with its segments dealt to workgroups by cost (R4R_GATHER_DEAL=1, experiment form: cost + deal as own launches)
R=${GRAFT_REPO_ROOT:-.}
line() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms']
print('%-28s %10.0f r/s %.4f ms  gemm %.4f gather %.4f' % ('$1', d['value'], d['ms_per_step'], k.get('proj_gemm_kernel', 0), k.get('proj_gather_max_kernel', 0)))"; }
for round in 1 2; do
  for w in "" "--workload cfg4_narre_kindle" "--workload cfg5_transnetpp_synthetic" "--doc-fill full --token-dist uniform"; do
    echo "== $w"
    python $R/bench.py --no-cpu-baseline $w 2>/dev/null | line "base"
    R4R_GATHER_DEAL=1 python $R/bench.py --no-cpu-baseline $w 2>/dev/null | line "dealt"
  done
done
