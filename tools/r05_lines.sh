#!/bin/bash
# the round's bench lines (profiles/r05_bench20_line.json: the driver's command; r05_bench_line.json: the default command)
O=gpurun_out/r05_lines; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench20_line.json 2> $O/bench20.err; tail -4 $O/bench20.err
( time python bench.py ) > $O/bench_line.json 2> $O/bench.err; tail -4 $O/bench.err
python3 - <<'PY'
import json
for f in ('bench20_line', 'bench_line'):
    d = json.loads(open('gpurun_out/r05_lines/%s.json' % f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], 'warmup', d['warmup'], 'roofline', d['roofline']['frac'], d['roofline'].get('steady_frac'), 'traffic', d['roofline']['traffic'])
    g = d['roofline_gather']; print('  gather', g['achieved'], g['frac'], g['traffic'], g.get('hbm_side_GBs'))
    for l in d.get('configs', []):
        if 'error' in l:
            print('  ', l); continue
        r = l.get('roofline', {})
        print('  %-30s %11.0f r/s %7.4f ms %5.1fs %s frac %s ach %s traffic %s' % (l['leg'], l['ratings_per_s'], l['ms_per_step'], l['leg_wall_s'], r.get('kernel'), r.get('frac'), r.get('achieved'), r.get('traffic')))
        if 'roofline_gather' in l:
            g = l['roofline_gather']; print('       gather', g['bound'], g['achieved'], g['frac'], g.get('hbm_side_frac'), g.get('traffic'), g['avg_launch_ms'])
    print('  cpu', d.get('cpu_baseline', {}).get('value'))
PY
