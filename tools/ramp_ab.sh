#!/bin/bash
# The driver's command (--steps 20 --warmup 5) under different untimed ramps: how many steps until the clocks
# are where a long run has them.  bash tools/ramp_ab.sh > gpurun_out/ramp_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
  for ramp in 30 100 300 1000; do
    python $R/bench.py --gpus 1 --steps 20 --warmup 5 --ramp $ramp --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('ramp=$ramp round $round  %10.0f ratings/s  %8.4f ms/step  gpu %8.4f  gemm %s' % (d['value'], d['ms_per_step'], d.get('gpu_ms_per_step', 0), d.get('kernel_ms')))"
  done
done
