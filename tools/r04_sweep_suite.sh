#!/bin/bash
# The ID-table families (the ones whose step ends in mf_adam_kernel): bash tools/r04_sweep_suite.sh [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {   # label, bench.py arguments
    local label=$1; shift
    python $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | tail -1 | \
        python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-36s %12.0f ratings/s  %8.4f ms/step (gpu %.4f)' % ('$label', d['value'], d['ms_per_step'], d['gpu_ms_per_step']))"
}
for rep in $(seq ${1:-1}); do
run "cfg1 bias_only" --workload cfg1_bias_only_musical
run "cfg2 MF_dot" --workload cfg2_mfdot_electronics
run "cfg2 MF_dot batch 8192" --workload cfg2_mfdot_electronics --batch-per-gpu 8192
run "cfg5 TransNet++" --workload cfg5_transnetpp_synthetic
run "cfg2 shapes, MF (L=32)" --workload cfg2_mfdot_electronics --model-type MF --latent 32
run "cfg2 shapes, NeuMF (L=32)" --workload cfg2_mfdot_electronics --model-type NeuMF --latent 32
R4R_SWEEP_PERIOD=1 run "cfg2 MF_dot, dense sweep" --workload cfg2_mfdot_electronics
R4R_SWEEP_PERIOD=1 run "cfg5 TransNet++, dense sweep" --workload cfg5_transnetpp_synthetic
done
