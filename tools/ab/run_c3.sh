#!/bin/bash
# A/B of library builds on the C3 workload (NeutronNova, 32 steps): tools/ab/run_c3.sh <outdir> <steps> <reps> name...
out=gpurun_out/$1; steps=$2; reps=$3; shift 3
mkdir -p $out
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    cp tools/ab/$v/*.so spartan2_amd/lib/
    python bench.py --workload c3 --steps $steps --warmup 3 > $out/$v.$rep.json 2> $out/$v.$rep.err
  done
done
