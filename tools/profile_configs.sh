#!/bin/bash
# rocprofv3 kernel traces (--kernel-trace --stats) of the bench loops of configs 3, 4 and 5 - the per-kernel evidence the config-2 job (profile_job.sh)
# gives for the headline. Usage (GPU box, repo root): tools/profile_configs.sh r06 -> gpurun_out/<round>_c{3,4,5}_kernel_stats_rocprof.csv
RD=${1:-r06}
R=$(pwd); O=$R/gpurun_out/profc_$RD; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o run -- python $R/bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline > $O/c3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -o run -- python $R/bench.py --workload c4 --steps 3 --warmup 1 --no-sharded --no-cpu-baseline > $O/c4.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o run -- python $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $O/c5.log 2>&1
cd $R
for c in c3 c4 c5; do
  f=$(find $O/$c -name "run_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/${RD}_${c}_kernel_stats_rocprof.csv
  tail -n 2 $O/$c.log | cut -c1-300
done
rm -rf $O
