#!/bin/bash
# The three rocprofv3 passes behind profiles/<tag>_kernel_stats.md and profiles/r01_pmc_traffic.json (run on the GPU box from the repo root):
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of their own (counters are never combined with other trace domains).
R=$(pwd); O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- $CMD > $O/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $CMD > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $CMD > $O/write.log 2>&1
cd $R
find $O -name "*.csv" | head -20
for d in kt fetch write; do f=$(find $O/$d -name "run_*" -name "*.csv" | head -1); dd=$(dirname "$f"); echo "$d -> $dd"; done
