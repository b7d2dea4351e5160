#!/bin/bash
# bench the prove time for several sizes of the resident sum-check tail (SPARTAN_TAIL_LOG2; 0 = off)
for lg in 13 14 15 16 15 16; do
  SPARTAN_TAIL_LOG2=$lg timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --concurrent 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail_log2=$lg', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms'].items()})"
done
