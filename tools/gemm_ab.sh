#!/bin/bash
# A/B of the projection GEMM's two forms on the bench workloads: bash tools/gemm_ab.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {
    local label=$1; shift
    "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-34s %9.0f ratings/s %8.4f ms/step  gemm %.4f ms  gather %.4f ms  rows %s' % ('$label', d['value'], d['ms_per_step'],
      d['kernel_ms'].get('proj_gemm_kernel', 0), d['kernel_ms'].get('proj_gather_max_kernel', 0), d.get('roofline', {}).get('distinct_token_rows_per_launch', '-')))"
}
for rep in 1 2; do
  run "balanced $*" python $R/bench.py --no-cpu-baseline "$@"
  run "tile     $*" env R4R_GEMM=tile python $R/bench.py --no-cpu-baseline "$@"
done
