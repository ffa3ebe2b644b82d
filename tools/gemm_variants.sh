#!/bin/bash
# A/B of projection-GEMM build variants (csrc/project.hip's R4R_* switches).
#   bash tools/gemm_variants.sh build            (in the build container: hipcc cross-compiles)
#   bash tools/gemm_variants.sh run [bench args] (on the MI355X box: two interleaved rounds of bench.py per variant)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/reviews4rec_amd/csrc
declare -A V=(
  [r2]="-DR4R_STG_PERM=0 -DR4R_EPI=0"
  [perm]="-DR4R_STG_PERM=1 -DR4R_EPI=0"
  [epi1]="-DR4R_STG_PERM=1 -DR4R_EPI=1"
  [epi1nt]="-DR4R_STG_PERM=1 -DR4R_EPI=1 -DR4R_EPI_NT=1"
  [nostore]="-DR4R_STG_PERM=1 -DR4R_EPI=2"
  [prio]="-DR4R_STG_PERM=1 -DR4R_EPI=1 -DR4R_PRIO=1"
)
ORDER=${ORDER:-"r2 perm epi1 epi1nt prio nostore"}
if [ "$1" = build ]; then
  for t in $ORDER; do make -s -C $C variant TAG=$t EXTRA="${V[$t]}" || exit 1; done
  exit 0
fi
shift
for rep in 1 2; do
  for t in $ORDER; do
    R4R_LIBRARY=$C/libr4r_hip_var_$t.so python $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms']
print('%-8s %9.0f ratings/s %8.4f ms/step  gemm %.4f  gather %.4f  head %.4f  bwd %.4f  reduce %.4f' % ('$t', d['value'], d['ms_per_step'],
      k.get('proj_gemm_kernel', 0), k.get('proj_gather_max_kernel', 0), k.get('deepconn_head_kernel', 0), k.get('deepconn_backward_kernel', 0), k.get('deepconn_reduce_kernel', 0)))"
  done
done
