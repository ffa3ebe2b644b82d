cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04b/pytest.log
tail -5 gpurun_out/r04b/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench20.json 2>gpurun_out/r04b/bench20.err; tail -c 1500 gpurun_out/r04b/bench20.json
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04b/bench200.json 2>gpurun_out/r04b/bench200.err
STEPS="--steps 200 --warmup 20" bash tools/prof_bench.sh r04b > gpurun_out/r04b/prof.log 2>&1
tail -12 gpurun_out/r04b/prof.log
