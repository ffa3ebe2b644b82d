#!/bin/bash
# round 5, first GPU contact: the driver's bench command with the new configuration legs + the bare two-rank command
O=gpurun_out/r05a; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench20.json 2> $O/bench20.err
tail -c 600 $O/bench20.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r05a/bench20.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], 'warmup', d['warmup'], 'roofline', d['roofline']['frac'], d['roofline'].get('steady_frac'))
print('gather', {k: d['roofline_gather'].get(k) for k in ('achieved', 'frac', 'walked_positions_per_launch', 'positions_per_launch', 'avg_launch_ms')})
for l in d.get('configs', []):
    if 'error' in l:
        print(l); continue
    r = l.get('roofline', {})
    print('%-32s %12.0f r/s %8.4f ms wall %5.1fs  %s frac %s ach %s' % (l['leg'], l['ratings_per_s'], l['ms_per_step'], l['leg_wall_s'], r.get('kernel'), r.get('frac'), r.get('achieved')))
    if 'roofline_gather' in l:
        g = l['roofline_gather']; print('     gather', g['bound'], g['achieved'], g['frac'], g.get('hbm_side_frac'), g['walked_positions_per_launch'], g['avg_launch_ms'])
print('cpu', d.get('cpu_baseline'))
PY
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -k "bare_two_gpu or more_ranks_than" 2>&1 | tail -15
