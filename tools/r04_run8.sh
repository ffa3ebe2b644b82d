cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench_plan.py tests/test_gpu_models.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -4
bash tools/r04_ab.sh "head4 base" 2 > gpurun_out/r04h_head.txt 2>&1; cat gpurun_out/r04h_head.txt
python tools/head_trace.py cfg3_deepconn_electronics_e300 2>&1 | tail -6
python tools/head_trace.py cfg3_deepconn_electronics_e300 --backward 2>&1 | tail -7
