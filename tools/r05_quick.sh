#!/bin/bash
# quick check of a build: the DeepCoNN engine's parity tests, then the default bench without the CPU leg
R=${GRAFT_REPO_ROOT:-.}
timeout 1200 python -m pytest $R/tests/test_gpu_engine.py $R/tests/test_gpu_bench_plan.py $R/tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
for i in 1 2; do
python $R/bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_ms']
print('%10.0f r/s %.4f ms (gpu %.4f)  ' % (d['value'], d['ms_per_step'], d['gpu_ms_per_step']), {a: round(b, 4) for a, b in k.items()})"
done
