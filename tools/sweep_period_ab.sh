#!/bin/bash
# cfg5 (TransNet++) and cfg2 (MF_dot): the temporally blocked table sweep at visit periods 1 (the dense sweep) .. 8.
# bash tools/sweep_period_ab.sh > gpurun_out/sweep_period_ab.txt   (WORKLOADS="..." picks the workloads)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for w in ${WORKLOADS:-cfg5_transnetpp_synthetic cfg2_mfdot_electronics}; do
for round in 1 2; do
  for period in 1 2 3 4 6 8; do
    R4R_SWEEP_PERIOD=$period python $R/bench.py --workload $w --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 | \
      python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$w period=$period round $round  %10.0f ratings/s  %8.4f ms/step  gpu %8.4f  %s' % (d['value'], d['ms_per_step'], d.get('gpu_ms_per_step', 0), d.get('kernel_ms')))"
  done
done
done
