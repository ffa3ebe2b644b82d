#!/bin/bash
# an epoch through the host loop (loader + native engine + validation pass), the record kept under profiles/
for mt in deepconn NARRE MF_dot; do
  python tools/bench_epoch.py --model-type $mt 2>/dev/null | tail -1
done
