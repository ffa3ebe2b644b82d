#!/bin/bash
# Round-4 record on the final tree: kernel-trace + PMC passes of every BASELINE config, the driver's 20-step line, the
# configuration table, the data-dependence points, the HBM-bound gather point, the NARRE strong-scaling legs, the
# one-rank RCCL legs.   bash tools/r04_final.sh   (on the MI355X box) -> gpurun_out/r04_final/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench20_line.json 2>/dev/null
python bench.py > $O/bench_line.json 2>/dev/null
STEPS="--steps 200 --warmup 20" bash tools/prof_bench.sh r04_bench > $O/prof_bench.log 2>&1
for w in cfg2_mfdot_electronics cfg4_narre_kindle cfg5_transnetpp_synthetic; do
  t=${w%%_*}
  STEPS="--steps 200 --warmup 20" BENCH_ARGS="--workload $w" bash tools/prof_bench.sh r04_bench_$t > $O/prof_$t.log 2>&1
done
STEPS="--steps 100 --warmup 20" BENCH_ARGS="--workload cfg5_transnetpp_synthetic --doc-fill full --token-dist uniform --conv-algo project" bash tools/prof_bench.sh r04_cfg5_fullunif > $O/prof_cfg5_fullunif.log 2>&1
cd $R
bash tools/bench_all.sh > $O/bench_all.txt 2>&1
bash tools/bench_points.sh > $O/bench_points.txt 2>&1
bash tools/r04_narre_strong.sh > $O/narre_strong.log 2>&1
bash tools/dp1_bench.sh --workload cfg2_mfdot_electronics > $O/rccl1_cfg2.log 2>&1
bash tools/dp1_bench.sh --workload cfg5_transnetpp_synthetic > $O/rccl1_cfg5.log 2>&1
bash tools/dp1_bench.sh > $O/rccl1_cfg3.log 2>&1
for t in r04_bench r04_bench_cfg2 r04_bench_cfg4 r04_bench_cfg5 r04_cfg5_fullunif; do
  cp gpurun_out/prof_$t/kernel_stats.csv $O/${t}_kernel_stats.csv; cp gpurun_out/prof_$t/pmc_summary.json $O/${t}_pmc_summary.json
done
bash tools/r04_sweep_suite.sh 2 > $O/sweep_suite.txt 2>&1
./scratch/dispatch_rate > $O/dispatch_rate.txt 2>&1
for w in cfg2_mfdot_electronics cfg5_transnetpp_synthetic; do python tools/sweep_trace.py $w 2>/dev/null | tail -12 > $O/sweep_trace_${w%%_*}.txt; done
ls -la $O
