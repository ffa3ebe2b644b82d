#!/bin/bash
# Sweep-chunk size A/B (build variants with -DR4R_MF_CHUNK=...): bash tools/chunk_ab.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { # label, env..., args
  local label=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-40s %10.0f ratings/s  %8.4f ms/step  %s' % ('$label', d['value'], d['ms_per_step'], d.get('kernel_ms')))"
}
for round in 1 2; do
for c in 8192 4096 2048; do
  L=$R/reviews4rec_amd/csrc/libr4r_hip_var_c$c.so; [ $c = 8192 ] && L=$R/reviews4rec_amd/csrc/libr4r_hip.so
  run "chunk=$c cfg5 period8" R4R_LIBRARY=$L python $R/bench.py --workload cfg5_transnetpp_synthetic --no-cpu-baseline
  run "chunk=$c cfg2 period8" R4R_LIBRARY=$L python $R/bench.py --workload cfg2_mfdot_electronics --no-cpu-baseline
  run "chunk=$c cfg5 period1" R4R_LIBRARY=$L R4R_SWEEP_PERIOD=1 python $R/bench.py --workload cfg5_transnetpp_synthetic --no-cpu-baseline
  run "chunk=$c cfg2 period1" R4R_LIBRARY=$L R4R_SWEEP_PERIOD=1 python $R/bench.py --workload cfg2_mfdot_electronics --no-cpu-baseline
  run "chunk=$c cfg2 B8192" R4R_LIBRARY=$L python $R/bench.py --workload cfg2_mfdot_electronics --batch-per-gpu 8192 --no-cpu-baseline
done
done
