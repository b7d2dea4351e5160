#!/bin/bash
# Kernel trace of a short C2 bench (headline + reference-order legs only); the stats table goes to gpurun_out/r03_quick_stats.csv
R=$(pwd); O=$R/gpurun_out/qprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0 --no-sharded ${EXTRA_BENCH_ARGS} > $O/log.txt 2>&1
cd $R
f=$(find $O -name "run_kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r03_quick_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03_quick_stats.csv')))
for r in rows[:45]:
    n=r['Name'][:70]
    if 'table' in n or 'comb_' in n or 'jac_to_affine' in n: continue
    print(f"{n:72s} {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} min {float(r['MinNs'])/1e3:9.1f}")
PY
rm -rf $O
