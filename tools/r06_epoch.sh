#!/bin/bash
# an epoch through the host loop (loader + native engine + validation pass), spans on and off, the engine on the
# loader's own batches held resident beside it; the record kept under profiles/r06_epoch.txt
for mt in deepconn NARRE MF_dot; do
  e=300; [ $mt = NARRE ] && e=64
  for sp in 1 0; do
    python tools/bench_epoch.py --model-type $mt --embed $e --spans $sp 2>/dev/null | tail -1
  done
done
