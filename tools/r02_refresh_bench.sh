#!/bin/bash
# Refreshes the committed bench lines of round 2 (profiles/r02_bench*.json) from the current build; run on the GPU box from the repo root.
set -x
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
python bench.py --workload c3 --steps 20 --warmup 3 > gpurun_out/r02_bench_c3.json 2> gpurun_out/r02_bench_c3.err
python bench.py --workload c4 --steps 5 --warmup 1 --no-sharded > gpurun_out/r02_bench_c4.json 2> gpurun_out/r02_bench_c4.err
python bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/r02_bench_c5.json 2> gpurun_out/r02_bench_c5.err
for f in "" _c3 _c4 _c5; do python -c "
import json;b=json.load(open('gpurun_out/r02_bench$f.json'));print('$f', round(b['ms_per_step'],3), round(b['value']/1e6,1), (b.get('cpu_baseline') or {}).get('gpu_proof_bit_exact_and_verified'))"; done
