#!/bin/bash
# gather with hot projected rows staged in LDS (experiment): A/B on frequency-ranked token ids
R=${GRAFT_REPO_ROOT:-.}; C=$R/reviews4rec_amd/csrc
line() { python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms']
print('%-28s %10.0f r/s %.4f ms  gemm %.4f gather %.4f' % ('$1', d['value'], d['ms_per_step'], k.get('proj_gemm_kernel', 0), k.get('proj_gather_max_kernel', 0)))"; }
for round in 1 2; do
  for w in "" "--workload cfg4_narre_kindle"; do
    echo "== $w"
    python $R/bench.py --no-cpu-baseline $w 2>/dev/null | line "base permuted"
    R4R_SYNTH_TOKEN_ORDER=rank python $R/bench.py --no-cpu-baseline $w 2>/dev/null | line "base ranked"
    for t in $TAGS; do
      R4R_SYNTH_TOKEN_ORDER=rank R4R_LIBRARY=$C/libr4r_hip_var_$t.so python $R/bench.py --no-cpu-baseline $w 2>/dev/null | line "$t ranked"
    done
  done
done
