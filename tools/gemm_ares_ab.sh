#!/bin/bash
# A/B of the projection GEMM's forms and of the A-resident form's knobs (pass split R1 / R2 / rest, chunks per
# barrier S, cache bits of the early / last stores).
#   bash tools/gemm_ares_ab.sh build | run [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/reviews4rec_amd/csrc
declare -A V=(
  [a43s2]="-DR4R_AR_R1=4 -DR4R_AR_R2=3 -DR4R_AR_S=2"
  [a43s1]="-DR4R_AR_R1=4 -DR4R_AR_R2=3 -DR4R_AR_S=1"
  [a43s4]="-DR4R_AR_R1=4 -DR4R_AR_R2=3 -DR4R_AR_S=4"
  [a43pl]="-DR4R_AR_R1=4 -DR4R_AR_R2=3 -DR4R_AR_S=2 -DR4R_AR_AUX=0"
  [a43wt]="-DR4R_AR_R1=4 -DR4R_AR_R2=3 -DR4R_AR_S=2 -DR4R_AR_AUX_LAST=16"
  [a34s2]="-DR4R_AR_R1=3 -DR4R_AR_R2=4 -DR4R_AR_S=2"
  [a322]="-DR4R_AR_R1=3 -DR4R_AR_R2=2 -DR4R_AR_S=2"
  [a232]="-DR4R_AR_R1=2 -DR4R_AR_R2=3 -DR4R_AR_S=2"
  [a331]="-DR4R_AR_R1=3 -DR4R_AR_R2=3 -DR4R_AR_S=2"
  [a52]="-DR4R_AR_R1=5 -DR4R_AR_R2=2 -DR4R_AR_S=2 -DR4R_AR_AUX=0"
  [a61]="-DR4R_AR_R1=6 -DR4R_AR_R2=1 -DR4R_AR_S=2 -DR4R_AR_AUX=0"
  [a43nt]="-DR4R_AR_R1=4 -DR4R_AR_R2=3 -DR4R_AR_S=2 -DR4R_AR_AUX=2"
  [a52s1]="-DR4R_AR_R1=5 -DR4R_AR_R2=2 -DR4R_AR_S=1 -DR4R_AR_AUX=0"
  [a52ns]="-DR4R_AR_R1=5 -DR4R_AR_R2=2 -DR4R_AR_S=2 -DR4R_AR_AUX=0 -DR4R_EPI=2"
  [db1]="-DR4R_AR_DB=1"
  [db3]="-DR4R_AR_DB=3"
  [db3ns]="-DR4R_AR_DB=3 -DR4R_EPI=2"
  [c2]="-DR4R_AR_SPLIT=2"
  [c2s1]="-DR4R_AR_SPLIT=2 -DR4R_AR_S=1"
  [c2ns]="-DR4R_AR_SPLIT=2 -DR4R_EPI=2"
  [cols]="-DR4R_AR_SPLIT=1"
  [colss1]="-DR4R_AR_SPLIT=1 -DR4R_AR_S=1"
  [colsns]="-DR4R_AR_SPLIT=1 -DR4R_EPI=2"
  [a43ns]="-DR4R_AR_R1=4 -DR4R_AR_R2=3 -DR4R_AR_S=2 -DR4R_EPI=2"
)
ORDER=${ORDER:-"a43s2 a43s1 a43s4 a43pl a43wt a34s2 a322 a232 a331 a43ns"}
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
