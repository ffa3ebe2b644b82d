for fill in lognormal full; do for dist in zipf uniform; do
python bench.py --no-cpu-baseline --workload cfg5_transnetpp_synthetic --doc-fill $fill --token-dist $dist --conv-algo project 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
g = d.get('roofline_gemm', d.get('roofline', {}))
print('$fill/$dist', d['ms_per_step'], 'gemm', g.get('avg_launch_ms'), 'rows', g.get('distinct_token_rows_per_launch'), 'gather', d['kernel_ms'].get('proj_gather_max_kernel'))"
done; done
