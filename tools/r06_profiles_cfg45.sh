#!/bin/bash
R=$GRAFT_REPO_ROOT
run() { local tag=$1; shift; BENCH_ARGS="$*" bash $R/tools/prof_bench.sh r06_$tag > $R/gpurun_out/prof_r06_$tag.log 2>&1; cp $R/gpurun_out/prof_r06_$tag/kernel_stats.csv $R/gpurun_out/r06_${tag}_kernel_stats.csv; cp $R/gpurun_out/prof_r06_$tag/pmc_summary.json $R/gpurun_out/r06_${tag}_pmc_summary.json; head -7 $R/gpurun_out/prof_r06_$tag/kernel_stats.csv | cut -c1-100; }
run bench_cfg5 --workload cfg5_transnetpp_synthetic
run cfg5_fullunif --workload cfg5_transnetpp_synthetic --doc-fill full --token-dist uniform --conv-algo project
run bench_cfg4 --workload cfg4_narre_kindle
