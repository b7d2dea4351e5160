#!/bin/bash
# Round 3: rocprofv3 kernel trace + the two PMC passes of the default bench command (separate runs), summaries under gpurun_out/ (copied to profiles/ by hand).
set -x
bash tools/profile_round.sh > gpurun_out/r03_profile.log 2>&1
KT=$(dirname $(find gpurun_out/prof/kt -name "run_kernel_trace.csv" | head -1))
FE=$(dirname $(find gpurun_out/prof/fetch -name "run_counter_collection.csv" | head -1))
WR=$(dirname $(find gpurun_out/prof/write -name "run_counter_collection.csv" | head -1))
python tools/summarize_profile.py $KT gpurun_out/r03_kernel_stats.md --title "round 3: rocprofv3 --kernel-trace --stats of bench.py (C2 prove, reference-order prove, sharded C4 legs)" --command "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0" --pmc-fetch $FE --pmc-write $WR --pmc-json gpurun_out/r03_pmc_traffic.json >> gpurun_out/r03_profile.log 2>&1
cp $KT/run_kernel_stats.csv gpurun_out/r03_kernel_stats_rocprof.csv 2>/dev/null
rm -rf gpurun_out/prof
tail -3 gpurun_out/r03_profile.log
