#!/bin/bash
# the round's bench lines (profiles/r06_bench20_line.json: the driver's command; r06_bench_line.json: the default command),
# each with its full record (what bench.py writes to gpurun_out/bench_line_full.json and stderr)
O=gpurun_out/r06_lines; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench20_line.json 2> $O/bench20.err; tail -4 $O/bench20.err | cut -c1-200
cp gpurun_out/bench_line_full.json $O/bench20_full.json
( time python bench.py ) > $O/bench_line.json 2> $O/bench.err; tail -4 $O/bench.err | cut -c1-200
cp gpurun_out/bench_line_full.json $O/bench_full.json
wc -c $O/*.json
python3 - <<'PY'
import json
for f in ('bench20_line', 'bench_line'):
    d = json.loads(open('gpurun_out/r06_lines/%s.json' % f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], 'warmup', d['warmup'], d['warmup_effective'], 'steady', d.get('steady', {}).get('ratings_per_s'),
          'roofline', d['roofline']['frac'], d['roofline'].get('steady_frac'), 'traffic', d['roofline']['traffic'])
    for k, v in d.get('legs', {}).items():
        print('   %-32s %s' % (k, v))
    print('   opt-in', d.get('legs_opt_in_f16_split'), 'host loop', d.get('host_loop'), 'cpu', d.get('cpu_baseline', {}).get('value'))
PY
