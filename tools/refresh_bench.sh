#!/bin/bash
# Refreshes the bench lines of a round (gpurun_out/<round>_bench*.json, copied into profiles/ by hand) from the current build; run on the GPU box
# from the repo root: tools/refresh_bench.sh r05
RD=${1:-r05}
set -x
python bench.py > gpurun_out/${RD}_bench.json 2> gpurun_out/${RD}_bench.err
python bench.py --workload c3 --steps 20 --warmup 3 > gpurun_out/${RD}_bench_c3.json 2> gpurun_out/${RD}_bench_c3.err
python bench.py --workload c4 --steps 5 --warmup 1 --no-sharded > gpurun_out/${RD}_bench_c4.json 2> gpurun_out/${RD}_bench_c4.err
python bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/${RD}_bench_c5.json 2> gpurun_out/${RD}_bench_c5.err
for m in 1024 4096 8192; do python bench.py --message-bytes $m --no-sharded --concurrent 0 > gpurun_out/${RD}_bench_${m}B.json 2> gpurun_out/${RD}_bench_${m}B.err; done
for f in "" _c3 _c4 _c5 _1024B _4096B _8192B; do python -c "
import json;b=json.load(open('gpurun_out/${RD}_bench$f.json'));print('$f', round(b['ms_per_step'],3), round(b['value']/1e6,1), (b.get('cpu_baseline') or {}).get('gpu_proof_bit_exact_and_verified'), (b.get('reference_order') or {}).get('ms_per_step'))"; done
