#!/bin/bash
# A/B on one box: the A-resident GEMM with / without the half-tile shares (R4R_AR_HALF), device durations per launch
# over the default bench command (8 pool batches: three of them have more than 7 1/3 row tiles per workgroup).
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in default nohalf; do
    if [ $v = default ]; then unset R4R_LIBRARY; else export R4R_LIBRARY=$R/reviews4rec_amd/csrc/libr4r_hip_var_$v.so; fi
    STEPS="--steps 200 --warmup 20" bash $R/tools/fence_probe.sh ab_$v > /tmp/ab.txt 2>&1
    echo "== $v: $(grep -o '"value": [0-9.]*' $R/gpurun_out/fence_ab_$v/bench.log | head -1) $(grep -o '"gpu_ms_per_step": [0-9.]*' $R/gpurun_out/fence_ab_$v/bench.log | head -1)"
    grep "gemm dur" /tmp/ab.txt | tail -1 | awk '{n=NF; for(i=n-31;i<=n;i++) printf "%s ", $i; print ""}'
  done
done
