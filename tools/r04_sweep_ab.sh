#!/bin/bash
# A/B of sweep build variants (libr4r_hip_var_<tag>.so built by `make variant UNIT=mf_engine.hip`): interleaved rounds
#   bash tools/r04_sweep_ab.sh "<tags>" [rounds] [bench args...]      ('base' = the in-tree library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/reviews4rec_amd/csrc
TAGS=$1; ROUNDS=${2:-2}; shift; shift
for rep in $(seq $ROUNDS); do
  for t in $TAGS; do
    L=$C/libr4r_hip_var_$t.so; [ $t = base ] && L=$C/libr4r_hip.so
    R4R_LIBRARY=$L python $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-10s %9.0f ratings/s %8.4f ms/step (gpu %.4f)' % ('$t', d['value'], d['ms_per_step'], d['gpu_ms_per_step']))"
  done
done
