#!/bin/bash
# where the host loop's time goes (tools/host_probe.py): the loader's iterator, train_step, main.train per batch and through spans
mkdir -p gpurun_out
for m in deepconn NARRE MF_dot; do
  e=300; [ $m = NARRE ] && e=64
  echo "== host_probe $m" ; timeout 300 python tools/host_probe.py --model-type $m --embed $e 2>&1 | grep -v amdgpu.ids | head -12
done > gpurun_out/r06_host_probe_after.txt 2>&1
cat gpurun_out/r06_host_probe_after.txt
