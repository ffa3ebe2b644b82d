#!/bin/bash
# The round's profile set (profiles/r06_*): kernel-trace stats + the four PMC passes of the default bench command and of
# every configuration leg of the bench line.  On the MI355X box: bash tools/r06_profiles.sh
R=$GRAFT_REPO_ROOT
run() {   # tag, bench args
    local tag=$1; shift
    BENCH_ARGS="$*" bash $R/tools/prof_bench.sh r06_$tag > $R/gpurun_out/prof_r06_$tag.log 2>&1
    cp $R/gpurun_out/prof_r06_$tag/kernel_stats.csv $R/gpurun_out/r06_${tag}_kernel_stats.csv
    cp $R/gpurun_out/prof_r06_$tag/pmc_summary.json $R/gpurun_out/r06_${tag}_pmc_summary.json
    head -7 $R/gpurun_out/prof_r06_$tag/kernel_stats.csv
}
run bench
run bench_cfg1 --workload cfg1_bias_only_musical
run bench_cfg2 --workload cfg2_mfdot_electronics
run bench_cfg2_b8192 --workload cfg2_mfdot_electronics --batch-per-gpu 8192
run bench_cfg4 --workload cfg4_narre_kindle
run bench_cfg5 --workload cfg5_transnetpp_synthetic
run bench_cfg3_fullunif --doc-fill full --token-dist uniform
run cfg5_fullunif --workload cfg5_transnetpp_synthetic --doc-fill full --token-dist uniform --conv-algo project
