#!/bin/bash
# Round 6 baseline: where the host loop's time goes (tools/host_probe.py) and the epoch numbers before the span entry.
mkdir -p gpurun_out
for m in deepconn NARRE MF_dot; do
  e=300; [ $m = NARRE ] && e=64
  echo "== host_probe $m" ; timeout 300 python tools/host_probe.py --model-type $m --embed $e 2>&1 | tail -40
done > gpurun_out/r06_host_probe_before.txt 2>&1
for m in deepconn NARRE MF_dot; do
  e=300; [ $m = NARRE ] && e=64
  timeout 300 python tools/bench_epoch.py --model-type $m --embed $e 2>&1 | tail -1
done > gpurun_out/r06_epoch_before.txt 2>&1
tail -5 gpurun_out/r06_epoch_before.txt
