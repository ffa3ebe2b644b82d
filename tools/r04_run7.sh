cd $GRAFT_REPO_ROOT
python tools/head_trace.py cfg3_deepconn_electronics_e300 > gpurun_out/r04g_head_trace.txt 2>&1
python tools/head_trace.py cfg3_deepconn_electronics_e300 --backward > gpurun_out/r04g_bwd_trace.txt 2>&1
python tools/head_trace.py cfg3_deepconn_electronics_e300 --gather > gpurun_out/r04g_gather_trace.txt 2>&1
tail -30 gpurun_out/r04g_head_trace.txt; tail -30 gpurun_out/r04g_bwd_trace.txt; tail -20 gpurun_out/r04g_gather_trace.txt
