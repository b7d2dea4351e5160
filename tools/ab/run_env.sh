#!/bin/bash
# A/B of environment switches of ONE build on one box, bench legs interleaved. Usage: tools/ab/run_env.sh <outdir> <steps> <reps> name=VAR=value ...
# (name "base" with an empty assignment runs the defaults: base=)
out=gpurun_out/$1; steps=$2; reps=$3; shift 3
mkdir -p $out
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    name=${spec%%=*}; assign=${spec#*=}
    if [ -n "$assign" ]; then
      env "$assign" python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-sharded --concurrent 0 > $out/$name.$rep.json 2> $out/$name.$rep.err
    else
      python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-sharded --concurrent 0 > $out/$name.$rep.json 2> $out/$name.$rep.err
    fi
  done
done
