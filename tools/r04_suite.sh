cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04s/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04s/pytest.log
tail -4 gpurun_out/r04s/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04s/bench20.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('gpurun_out/r04s/bench20.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['gpu_ms_per_step'],d['roofline']['frac'],d['roofline']['launches'],d['steady']['ratings_per_s'])"
