#!/bin/bash
# Round 3: 100 000 proofs x 8 contexts with 24 hardware queues (VERDICT r2 item 2's done-criterion), then 40 000 with the default 4.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r03_stress_100k.txt
: > $out
for q in 24 4; do
  n=12500; [ $q = 4 ] && n=5000
  echo "=== GPU_MAX_HW_QUEUES=$q, 8 contexts x $n proofs" >> $out
  GPU_MAX_HW_QUEUES=$q timeout 2400 python tools/concurrency_stress.py --contexts 8 --proofs $n >> $out 2>&1
  echo "rc=$?" >> $out
done
cat $out
