#!/bin/bash
# A/B of the projection GEMM's forms and of the A-resident form's remaining build knobs (chunks per barrier, weight
# prefetch depth, the no-store timing build).  The pass-structure variants of round 3 (profiles/r03a_gemm_variants.txt)
# were measured with earlier versions of this script and are no longer in the source.
#   bash tools/gemm_ares_ab.sh build | run [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/reviews4rec_amd/csrc
declare -A V=(
  [def]=""
  [s2]="-DR4R_AR_S=2"
  [db3]="-DR4R_AR_DB=3"
  [ns]="-DR4R_EPI=2"
)
ORDER=${ORDER:-"def s2 db3 ns"}
if [ "$1" = build ]; then
  for t in $ORDER; do make -s -C $C variant TAG=$t EXTRA="${V[$t]}" || exit 1; done
  exit 0
fi
shift
line() {
  python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms']
print('%-14s %9.0f ratings/s %8.4f ms/step (gpu %.4f)  gemm %.4f  gather %.4f' % ('$1', d['value'], d['ms_per_step'], d.get('gpu_ms_per_step', 0),
      k.get('proj_gemm_kernel', 0), k.get('proj_gather_max_kernel', 0)))"
}
for rep in 1 2; do
  R4R_GEMM=balanced python $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | tail -1 | line balanced
  for t in $ORDER; do
    R4R_GEMM=ares R4R_LIBRARY=$C/libr4r_hip_var_$t.so python $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | tail -1 | line ares-$t
  done
done
