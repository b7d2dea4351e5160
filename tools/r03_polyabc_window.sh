#!/bin/bash
# poly_ABC column order: global length sort (round 2) against length sort within windows of W consecutive columns; HIP-event time of the kernel class
# inside a config-2 prove (bench.py's instrumented pass) and the prove itself.
cd "$(dirname "$0")/.."
for w in 0 1024 4096 8192 32768 131072; do
  SPARTAN_POLYABC_WINDOW=$w python bench.py --steps 10 --warmup 2 --no-sharded --concurrent 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['other_kernels']
print('window $w: poly_abc', round(k['poly_abc']['avg_us'],1), 'us', round(k['poly_abc']['alg_GBps']), 'GB/s; spmv', round(k['spmv_incremental']['avg_us'],1), 'us; prove', round(d['ms_per_step'],4), 'ms; inner', round(d['phases_ms']['inner_sumcheck'],3))"
done
