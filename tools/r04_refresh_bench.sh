#!/bin/bash
# Refreshes the committed bench lines of round 4 (profiles/r04_bench*.json) from the current build; run on the GPU box from the repo root.
set -x
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
python bench.py --workload c3 --steps 20 --warmup 3 > gpurun_out/r04_bench_c3.json 2> gpurun_out/r04_bench_c3.err
python bench.py --workload c4 --steps 5 --warmup 1 --no-sharded > gpurun_out/r04_bench_c4.json 2> gpurun_out/r04_bench_c4.err
python bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/r04_bench_c5.json 2> gpurun_out/r04_bench_c5.err
for m in 1024 4096 8192; do python bench.py --message-bytes $m --no-sharded --concurrent 0 > gpurun_out/r04_bench_${m}B.json 2> gpurun_out/r04_bench_${m}B.err; done
for f in "" _c3 _c4 _c5 _1024B _4096B _8192B; do python -c "
import json;b=json.load(open('gpurun_out/r04_bench$f.json'));print('$f', round(b['ms_per_step'],3), round(b['value']/1e6,1), (b.get('cpu_baseline') or {}).get('gpu_proof_bit_exact_and_verified'), (b.get('reference_order') or {}).get('ms_per_step'))"; done
