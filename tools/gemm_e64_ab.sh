#!/bin/bash
# cfg4 / cfg5 (E = 64): the weight-resident GEMM (form 4, the default there; 512 and 1,024 threads) against the
# A-resident form (R4R_GEMM=ares), device durations from a kernel trace of the default bench command
R=$GRAFT_REPO_ROOT
for wl in cfg4_narre_kindle cfg5_transnetpp_synthetic; do
  for f in ${FORMS:-default ares}; do
    unset R4R_GEMM R4R_LIBRARY
    [ $f = ares ] && export R4R_GEMM=ares
    case $f in default|ares) ;; *) export R4R_LIBRARY=$R/reviews4rec_amd/csrc/libr4r_hip_var_$f.so;; esac
    STEPS="--steps 200 --warmup 20" BENCH_ARGS="--workload $wl" ROWS=0 bash $R/tools/fence_probe.sh e64_$f > /tmp/ab.txt 2>&1
    echo "== $wl $f: $(grep -o '"value": [0-9.]*' $R/gpurun_out/fence_e64_$f/bench.log | head -1)"
    grep "proj_gemm\|gather_max" $R/gpurun_out/fence_e64_$f/kernel_stats.csv | cut -c1-90
  done
done
