#!/bin/bash
# Every BASELINE.json configuration on one MI355X, native step vs the hipGraph-captured op-by-op
# step: bash tools/bench_all.sh > gpurun_out/bench_all.txt   (the table kept in profiles/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {   # label, bench.py arguments
    local label=$1; shift
    python $R/bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 | \
        python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-44s %12.0f ratings/s  %8.4f ms/step' % ('$label', d['value'], d['ms_per_step']))"
}
for w in cfg1_bias_only_musical cfg2_mfdot_electronics cfg3_deepconn_electronics_e300 cfg4_narre_kindle cfg5_transnetpp_synthetic; do
    run "$w native" --workload $w
    run "$w graph" --workload $w --engine graph
done
run "cfg2 MF_dot batch 8192 native" --workload cfg2_mfdot_electronics --batch-per-gpu 8192
run "cfg2 MF_dot batch 8192 graph" --workload cfg2_mfdot_electronics --batch-per-gpu 8192 --engine graph
run "cfg3 shapes, deepconn++ native" --model-type deepconn++
run "cfg3 shapes, deepconn++ graph" --model-type deepconn++ --engine graph
run "cfg3 E=64 native" --embed 64
run "cfg5 shapes, transnet native" --workload cfg5_transnetpp_synthetic --model-type transnet
run "cfg5 shapes, transnet graph" --workload cfg5_transnetpp_synthetic --model-type transnet --engine graph
run "cfg3 from host memory native" --from-host
run "cfg2 shapes, MF (L=32) native" --workload cfg2_mfdot_electronics --model-type MF --latent 32
run "cfg2 shapes, MF (L=32) graph" --workload cfg2_mfdot_electronics --model-type MF --latent 32 --engine graph
run "cfg2 shapes, NeuMF (L=32) native" --workload cfg2_mfdot_electronics --model-type NeuMF --latent 32
run "cfg2 shapes, NeuMF (L=32) graph" --workload cfg2_mfdot_electronics --model-type NeuMF --latent 32 --engine graph
run "cfg3 fp16-split GEMM (opt-in) native" --gemm-math f16x2
run "cfg4 fp16-split GEMM (opt-in) native" --workload cfg4_narre_kindle --gemm-math f16x2
run "cfg4 shapes, NARRE E=300 native" --workload cfg4_narre_kindle --embed 300
run "cfg4 shapes, NARRE E=300 graph" --workload cfg4_narre_kindle --embed 300 --engine graph
run "cfg3 batch 1024 native" --batch-per-gpu 1024
