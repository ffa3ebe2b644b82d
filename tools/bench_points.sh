#!/bin/bash
# The headline's data dependence: cfg3 at the SURVEY 8d default, with documents that
# fill all T positions, and with uniform word ids -- project-then-gather vs the direct conv at each point.
# bash tools/bench_points.sh > gpurun_out/bench_points.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {
    local label=$1; shift
    python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | \
        python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('roofline', {})
print('%-46s %10.0f ratings/s %8.4f ms/step  rows %7s  %s %.4f ms  gather %s ms' % ('$label', d['value'], d['ms_per_step'],
      r.get('distinct_token_rows_per_launch', '-'), r.get('kernel', '-'), r.get('avg_launch_ms', 0), d['kernel_ms'].get('proj_gather_max_kernel', '-')))"
}
for w in ${WORKLOADS:-cfg3_deepconn_electronics_e300}; do
  for fill in lognormal full; do
    for dist in zipf uniform; do
        run "$w $fill/$dist auto" --workload $w --doc-fill $fill --token-dist $dist
        run "$w $fill/$dist project" --workload $w --doc-fill $fill --token-dist $dist --conv-algo project
        run "$w $fill/$dist direct" --workload $w --doc-fill $fill --token-dist $dist --conv-algo direct
    done
  done
done
