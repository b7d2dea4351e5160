set -x
bash tools/profile_round.sh > gpurun_out/r02final_profile.log 2>&1
KT=$(dirname $(find gpurun_out/prof/kt -name "run_kernel_trace.csv" | head -1))
FE=$(dirname $(find gpurun_out/prof/fetch -name "run_counter_collection.csv" | head -1))
WR=$(dirname $(find gpurun_out/prof/write -name "run_counter_collection.csv" | head -1))
python tools/summarize_profile.py $KT gpurun_out/r02final_kernel_stats.md --title "round 2: rocprofv3 --kernel-trace --stats of bench.py (C2 prove + sharded C4 legs)" --command "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --concurrent 0" --pmc-fetch $FE --pmc-write $WR --pmc-json gpurun_out/r02final_pmc_traffic.json >> gpurun_out/r02final_profile.log 2>&1
cp $KT/run_kernel_stats.csv gpurun_out/r02final_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof
python bench.py > gpurun_out/r02final_bench.json 2> gpurun_out/r02final_bench.err
python bench.py --workload c3 --steps 10 --warmup 2 > gpurun_out/r02final_bench_c3.json 2> gpurun_out/r02final_bench_c3.err
tail -3 gpurun_out/r02final_profile.log; cat gpurun_out/r02final_bench.json | cut -c1-600; cat gpurun_out/r02final_bench_c3.json
