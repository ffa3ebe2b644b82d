cd $GRAFT_REPO_ROOT
python tools/head_trace.py cfg4_narre_kindle 2>&1 | tail -14
python tools/head_trace.py cfg4_narre_kindle --backward 2>&1 | tail -14
python bench.py --workload cfg4_narre_kindle --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4_narre_kindle --no-cpu-baseline > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) /tmp/ks.csv; head -8 /tmp/ks.csv
