R=${GRAFT_REPO_ROOT:-.}
for w in cfg5_transnetpp_synthetic cfg4_narre_kindle cfg2_mfdot_electronics; do
python $R/bench.py --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w %10.0f r/s %.4f ms' % (d['value'], d['ms_per_step']))"
done
python $R/bench.py --no-cpu-baseline --workload cfg2_mfdot_electronics --model-type MF --latent 32 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MF L=32 %10.0f r/s %.4f ms' % (d['value'], d['ms_per_step']))"
